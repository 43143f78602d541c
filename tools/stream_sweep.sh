# developer probe: the default configuration per stream count - pipelined ms per step, blocking p50, what the front-end costs,
# and the plan-switch sizes (4: fused limb passes; 16: grouped limb convs for the C = 256 stage; 17: ragged tiles)
cd /root/repo
for n in 1 2 3 4 8 15 16 17 24 32 40 48 64 96 128; do
  python bench.py --streams $n --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('streams %3d  %-4s ms/step %.4f  chunks/s %8.0f  blocking p50 %.3f  vocoder alone %.3f  front-end cost %.3f' % ($n, d['config']['arith'], d['ms_per_step'], d['value'], d['p50_latency_ms'], r['vocoder_alone_ms'] or 0, r['frontend_cost_ms'] or 0))"
done
