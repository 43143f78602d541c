cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run24; rm -rf $O; mkdir -p $O
export CONAN_RB_NOPAIR=1
rocprofv3 --kernel-trace --output-format csv -d $O/blk -o run -- python3 tools/blocking_trace.py 64 > $O/blk.log 2>&1
python3 tools/marked_stats.py $O/blk/run_kernel_trace.csv 6 > $O/blk_stats.csv
rm -f $O/*/run_kernel_trace.csv
python3 - <<'PY'
import csv
for r in csv.DictReader(open("/root/repo/gpurun_out/r3_run24/blk_stats.csv")):
    print("%-70s n/step %5.1f avg %7.1f us min %7.1f max %7.1f" % (r["Name"][:70], float(r["CallsPerStep"]), float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
