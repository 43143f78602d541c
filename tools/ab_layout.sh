# A/B on one box: the multi-tile decoder launch with groups on one XCD (group-fastest, L2 hand-offs: CONAN_MEGA_LAYOUT unset)
# against member-fastest (member s of every group on XCD s: strip s's weights stay in that XCD's L2; agent-scope hand-offs),
# alternating runs; then one FETCH_SIZE pass per layout (HBM fetch of the decoder launch per step).
cd /tmp && export TMPDIR=/tmp
cd /root/repo
B="python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10"
P="import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(sys.argv[1], 'ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms']))"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "group-fastest "
  CONAN_MEGA_LAYOUT=m $B 2>/dev/null | python -c "$P" "member-fastest"
done
for L in g m; do
  export CONAN_MEGA_LAYOUT=$L
  python tools/stage_times.py 64 2>/dev/null | grep -E "decoder|pipelined" | sed "s/^/layout $L: /"
  O=gpurun_out/ab_layout_$L; rm -rf $O; mkdir -p $O
  CMD="python3 bench.py --workload b64 --steps 4 --warmup 3 --marks --no-cpu-baseline --no-b1 --no-other"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -o run -- $CMD > $O/pmc.log 2>&1
  python3 tools/summarize_pmc.py --cmd "$CMD" --steps 4 --out $O/pmc.json $O/pmc_*/run_counter_collection.csv | grep -E "HBM bytes|decoder_mega|emformer" | sed "s/^/layout $L: /"
done
