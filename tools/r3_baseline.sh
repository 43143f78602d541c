# round-3 baseline: default bench, stage times, and a blocking-step kernel trace with per-launch durations
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_base; rm -rf $O; mkdir -p $O
python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 tools/stage_times.py 64 > $O/stage_times.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 tools/blocking_trace.py 64 > $O/trace.log 2>&1
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv > $O/seq_all.txt 2>&1
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv "conv_mfma_kernel<32, 64" "conv_mfma_kernel<64, 64, 2, 2, 1, 32" > $O/seq_ups.txt 2>&1
rm -f $O/trace/run_kernel_trace.csv
tail -c 1500 $O/bench.json; cat $O/stage_times.txt; head -30 $O/seq_all.txt; cat $O/seq_ups.txt
