# A/B of two builds of the library on one box: conan_amd/libconan_hip_old.so (built from another commit, developer artefact) against
# the tree's libconan_hip.so, alternating runs of the default bench line
cd /root/repo
cp conan_amd/libconan_hip.so /tmp/lib_new.so; cp conan_amd/libconan_hip_old.so /tmp/lib_old.so
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10"
P="import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[1], 'ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
for i in 1 2 3; do
  cp /tmp/lib_old.so conan_amd/libconan_hip.so; $B 2>/dev/null | python -c "$P" "old"
  cp /tmp/lib_new.so conan_amd/libconan_hip.so; $B 2>/dev/null | python -c "$P" "new"
done
