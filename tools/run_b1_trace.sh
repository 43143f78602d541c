# developer tool: kernel trace of the B=1 workload (latency analysis)
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/b1_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 bench.py --workload b1 --steps 40 --warmup 5 --no-cpu-baseline > $O/log.txt 2>&1
tail -2 $O/log.txt
ls $O
