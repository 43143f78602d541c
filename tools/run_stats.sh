# developer tool: rocprofv3 kernel stats of a short default bench run; environment of the caller is inherited
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/stats_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b1 > $O/log.txt 2>&1
rm -f $O/run_kernel_trace.csv
head -14 $O/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
