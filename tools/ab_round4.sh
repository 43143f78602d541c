cd /root/repo
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 80 --warmup 15"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default   ms/step %.4f p50 %.3f' % (d['ms_per_step'], d['p50_latency_ms']))"
  CONAN_MEGA_BLK=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('blk       ms/step %.4f p50 %.3f' % (d['ms_per_step'], d['p50_latency_ms']))"
  CONAN_BENCH_COMM=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('comm      ms/step %.4f p50 %.3f gathers %s' % (d['ms_per_step'], d['p50_latency_ms'], d['ranks']['gathers']))"
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x 2>&1 | tail -3
[ -x tools/bin/rb_bench_st4 ] && RB_LIMB=1 tools/bin/rb_bench_st4 64 4 20   # (a developer build of tools/rb_bench.hip; not built by the repo)
