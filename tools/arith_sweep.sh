# developer probe: the AUTO arithmetic heuristic - blocking p50 and pipelined ms/step per stream count in both forms
cd /root/repo
for n in 1 4 8 16 32; do
  for a in f32 limb auto; do
    python bench.py --streams $n --arith $a --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('streams %3d %-5s -> %-4s ms/step %.4f  p50 %.3f' % ($n, '$a', d['config']['arith'], d['ms_per_step'], d['p50_latency_ms']))"
  done
done
