cd /root/repo
python3 tools/stage_times.py 64 2>&1 | tail -2
for i in 1 2; do python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4), round(d['p50_latency_ms'],3))"; done
