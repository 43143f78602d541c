cd /root/repo
run() { python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 6 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],4), round(d['p50_latency_ms'],3))"; }
run base
CONAN_MEGA_GS=16 CONAN_MEGA_GRID=256 run gs16
CONAN_MEGA_GS=4 CONAN_MEGA_GRID=64 run gs4
CONAN_MEGA_GS=4 CONAN_MEGA_GRID=128 run gs4_g128
