cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run11; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 0 > $O/bench.json 2> $O/bench.err
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv > $O/seq_all.txt 2>&1
rm -f $O/trace/run_kernel_trace.csv
tail -c 300 $O/bench.json | head -c 300; echo
head -24 $O/seq_all.txt
