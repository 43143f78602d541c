# A/B on one box: the Emformer's cluster size in pipelined steps (default: groups x cs <= 64 workgroups -> cs = 2 at 64 streams) against
# CONAN_EMF_CLUSTER=1 (one workgroup per group: fewer whole-CU workgroups, each for longer)
cd /root/repo
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10"
P="import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[1], 'ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "default (cs = 2 pipelined, 8 blocking)"
  CONAN_EMF_CLUSTER=1 $B 2>/dev/null | python -c "$P" "CONAN_EMF_CLUSTER=1               "
done
