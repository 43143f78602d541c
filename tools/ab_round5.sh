# A/B on one box: groups on one XCD (default) against the agent-scope protocol everywhere (CONAN_MEGA_NOL2=1), alternating runs
cd /root/repo
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('l2 groups   ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
  CONAN_MEGA_NOL2=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('agent scope ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
done
