cd /root/repo
python3 tools/timeline.py 64 40 2>&1 | tail -9
run() { python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --latency-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],4), round(d['p50_latency_ms'],3), d['step_time_stats']['p50_ms'], d.get('latency_b1',{}).get('p50_latency_ms'))"; }
run mega
CONAN_DEC_MEGA=0 run nomega
CONAN_RB_NOPAIR=1 run mega_nopair
CONAN_MEGA_GRID=64 run mega_grid64
