# developer tool: kernel trace of the default (pipelined) workload for timeline analysis
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/b64_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-b1 --latency-steps 1 ${TRACE_ARGS} > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
ls -la $O
