# Collect the round profiles on a MI355X box (run from the repo root through gpurun): GPU tests, default bench, rocprofv3
# kernel stats, separate PMC FETCH_SIZE / WRITE_SIZE passes and their summary.  Outputs under gpurun_out/r1_final/;
# the summaries to keep are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r1_final; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-b1 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-b1 --latency-steps 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-b1 --latency-steps 1 > $O/pmc_write.log 2>&1
python3 tools/summarize_pmc.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-b1 --latency-steps 1" $O/pmc_hbm.json > $O/pmc_hbm.txt
rm -f $O/pmc_fetch/run_kernel_trace.csv $O/pmc_write/run_kernel_trace.csv $O/stats/run_kernel_trace.csv
cat $O/pytest_gpu.txt; cut -c1-400 $O/bench.json; head -8 $O/stats/run_kernel_stats.csv; head -12 $O/pmc_hbm.txt
