# Collect the round profiles on a MI355X box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh [workload] [arith]      (default b64 auto; arith = auto | f32 | limb)
# rocprofv3 kernel stats of the default bench schedule, then separate PMC passes (one counter group per pass, with
# --kernel-trace only) of `bench.py --marks`, summarised by tools/summarize_pmc.py over the marked (timed) steps.
# Outputs under gpurun_out/${R}_prof_<tag>/; the summaries to keep are copied into profiles/ (${R}_<tag>_*), tag = workload for
# the library's default arithmetic, workload_<arith> for an explicit one (the name bench.py looks its PMC summary up under).
W=${1:-b64}
A=${2:-auto}
R=${ROUND:-r6}
STEPS=${STEPS:-4}
TAG=$W; if [ "$A" != "auto" ]; then TAG=${W}_$A; fi
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/${R}_prof_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload $W --arith $A --steps 30 --warmup 5 --no-cpu-baseline --no-b1 --no-other > $O/stats.log 2>&1
# the same kernels in blocking steps only (the schedule bench.py's roofline events are taken in), restricted to the dispatches
# between two profile marks (tools/blocking_trace.py: 6 timed steps; tools/marked_stats.py) - the per-utterance style pass runs
# the same conv_mfma instantiations as the upsamplers and must not be averaged into them
B=64; case $W in b1*) B=1;; b128*) B=128;; esac
rocprofv3 --kernel-trace --output-format csv -d $O/stats_blocking -o run -- python3 tools/blocking_trace.py $B $A > $O/stats_blocking.log 2>&1
python3 tools/marked_stats.py $O/stats_blocking/run_kernel_trace.csv 6 > $O/stats_blocking/run_kernel_stats.csv
CMD="python3 bench.py --workload $W --arith $A --steps $STEPS --warmup 3 --marks"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -o run -- $CMD > $O/pmc_$T.log 2>&1
done
python3 tools/summarize_pmc.py --cmd "$CMD" --steps $STEPS --out $O/pmc.json $O/pmc_*/run_counter_collection.csv > $O/pmc.txt
mkdir -p profiles
cp $O/stats/run_kernel_stats.csv profiles/${R}_${TAG}_kernel_stats.csv 2>/dev/null
cp $O/stats_blocking/run_kernel_stats.csv profiles/${R}_${TAG}_kernel_stats_blocking.csv 2>/dev/null
cp $O/pmc.json profiles/${R}_${TAG}_pmc.json; cp $O/pmc.txt profiles/${R}_${TAG}_pmc.txt
mkdir -p gpurun_out/profiles_${R}; cp profiles/${R}_${TAG}_kernel_stats.csv profiles/${R}_${TAG}_kernel_stats_blocking.csv profiles/${R}_${TAG}_pmc.json profiles/${R}_${TAG}_pmc.txt gpurun_out/profiles_${R}/
rm -f $O/*/run_kernel_trace.csv
head -8 $O/stats/run_kernel_stats.csv | cut -c1-160; head -20 $O/pmc.txt
