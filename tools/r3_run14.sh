cd /root/repo
run() { python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
run cs4
CONAN_EMF_CLUSTER=1 run cs1
CONAN_EMF_CLUSTER=2 run cs2
done
