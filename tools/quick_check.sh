cd /root/repo
python -m pytest tests/test_gpu_arith.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_configs.py -q -x 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 80 --warmup 15"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms/step %.4f p50 %.3f frac %.3f' % (d['ms_per_step'], d['p50_latency_ms'], d['roofline']['frac']))"
done
