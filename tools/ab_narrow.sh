# A/B on one box: narrow layers of the multi-tile decoder launch as K-split 16-column strips (default) against 64-column strips (CONAN_MEGA_NARROW=0)
cd /root/repo
python tools/mega_probe.py 64 2>&1 | grep -E "rowconv<1,1,4>|in all|alone" | head -20
CONAN_MEGA_NARROW=0 python tools/mega_probe.py 64 2>&1 | grep -E "in all|alone"
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10"
P="import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[1], 'ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "k-split narrow layers"
  CONAN_MEGA_NARROW=0 $B 2>/dev/null | python -c "$P" "64-column strips     "
done
python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -2
