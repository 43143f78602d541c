cd /root/repo
run() { python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],4))"; }
for sk in 0 1 2 3; do CONAN_SKIP_STAGE=$sk run skip$sk; done
CONAN_EMF_CLUSTER=1 run emf1
CONAN_EMF_CLUSTER=2 run emf2
CONAN_EMF_CLUSTER=8 run emf8
