# Collect the round profiles on a MI355X box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh [workload]      (default b64)
# rocprofv3 kernel stats of the default bench schedule, then separate PMC passes (one counter group per pass, with
# --kernel-trace only) of `bench.py --marks`, summarised by tools/summarize_pmc.py over the marked (timed) steps.
# Outputs under gpurun_out/${R}_prof_<workload>/; the summaries to keep are copied into profiles/ (${R}_<workload>_*).
W=${1:-b64}
R=${ROUND:-r3}
STEPS=${STEPS:-4}
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/${R}_prof_$W; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-b1 > $O/stats.log 2>&1
# the same kernels in blocking steps only (the schedule bench.py's roofline events are taken in), restricted to the dispatches
# between two profile marks (tools/blocking_trace.py: 6 timed steps; tools/marked_stats.py) - the per-utterance style pass runs
# the same conv_mfma instantiations as the upsamplers and must not be averaged into them
# (the arithmetic switch of the workload: bench.py sets it itself; the blocking trace takes it from the environment)
if [ "$W" = "b64_bf16x3" ]; then export CONAN_RB_LIMB=1; else unset CONAN_RB_LIMB; fi
rocprofv3 --kernel-trace --output-format csv -d $O/stats_blocking -o run -- python3 tools/blocking_trace.py 64 > $O/stats_blocking.log 2>&1
unset CONAN_RB_LIMB
python3 tools/marked_stats.py $O/stats_blocking/run_kernel_trace.csv 6 > $O/stats_blocking/run_kernel_stats.csv
CMD="python3 bench.py --workload $W --steps $STEPS --warmup 3 --marks"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -o run -- $CMD > $O/pmc_$T.log 2>&1
done
python3 tools/summarize_pmc.py --cmd "$CMD" --steps $STEPS --out $O/pmc.json $O/pmc_*/run_counter_collection.csv > $O/pmc.txt
mkdir -p profiles
cp $O/stats/run_kernel_stats.csv profiles/${R}_${W}_kernel_stats.csv 2>/dev/null
cp $O/stats_blocking/run_kernel_stats.csv profiles/${R}_${W}_kernel_stats_blocking.csv 2>/dev/null
cp $O/pmc.json profiles/${R}_${W}_pmc.json; cp $O/pmc.txt profiles/${R}_${W}_pmc.txt
mkdir -p gpurun_out/profiles_${R}; cp profiles/${R}_${W}_kernel_stats.csv profiles/${R}_${W}_kernel_stats_blocking.csv profiles/${R}_${W}_pmc.json profiles/${R}_${W}_pmc.txt gpurun_out/profiles_${R}/
rm -f $O/*/run_kernel_trace.csv
head -8 $O/stats/run_kernel_stats.csv | cut -c1-160; head -20 $O/pmc.txt
