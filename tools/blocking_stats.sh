# developer probe: rocprofv3 kernel stats of 6 blocking fused steps at 64 streams (between two profile marks)
A=${1:-auto}
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/bt_$A; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/blocking_trace.py 64 $A > $O/log.txt 2>&1
python3 tools/marked_stats.py $O/run_kernel_trace.csv 6 > $O/stats.csv
rm -f $O/run_kernel_trace.csv
cut -d, -f1-5 $O/stats.csv | cut -c1-110
