cd /tmp && export TMPDIR=/tmp
cd /root/repo
RB_LIMB=1 timeout 100 tools/bin/rb_bench_st 64 4 20 | grep "^limb"
RB_MERGE=1 RB_LIMB=1 timeout 100 tools/bin/rb_bench_st 64 4 20 | grep "^limb"
RB_LIMB=1 timeout 100 tools/bin/rb_bench_st 5 3 20 | grep "^limb"
python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
for i in 1 2 3; do python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d.get('p50_latency_ms'), d.get('step_time_stats',{}).get('p95_ms'))"; done
