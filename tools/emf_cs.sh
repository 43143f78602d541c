cd /root/repo
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 120 --warmup 20"
for rep in 1 2; do
for r in d 0 4 8 16; do
  if [ $r = d ]; then unset CONAN_RESERVE_CUS; else export CONAN_RESERVE_CUS=$r; fi
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('reserve $r  ms/step %.4f p50 %.3f p95int %.3f' % (d['ms_per_step'], d['p50_latency_ms'], d['step_time_stats']['p95_ms']))"
done; done
