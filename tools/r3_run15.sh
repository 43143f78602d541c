cd /root/repo
run() { python3 bench.py --workload $W --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$W $1', round(d['ms_per_step'],4), round(d['p50_latency_ms'],3))"; }
for W in b128s2 b128s2mem4; do
run default
CONAN_DEC_MEGA=0 run nomega
CONAN_RB_NOPAIR=1 run nopair
CONAN_DEC_MEGA=0 CONAN_RB_NOPAIR=1 run neither
done
