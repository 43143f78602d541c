cd /tmp && export TMPDIR=/tmp
cd /root/repo
for v in "X=1" "CONAN_EMF_NOHOLD=1" "X=1" "CONAN_EMF_NOHOLD=1" "X=1" "CONAN_EMF_NOHOLD=1"; do
env $v python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d.get('p50_latency_ms'),4), round(d.get('step_time_stats',{}).get('p95_ms',0),3))"
done
