# developer probe: per-kernel durations of 6 blocking fused steps at ONE stream, and the gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/bt_b1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/blocking_trace.py 1 auto > $O/log.txt 2>&1
python3 tools/marked_stats.py $O/run_kernel_trace.csv 6 > $O/stats.csv
python3 - <<'PY'
import csv
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open("gpurun_out/bt_b1/run_kernel_trace.csv"))]
rows.sort()
marks=[i for i,r in enumerate(rows) if "profile_mark" in r[2]]
seg=rows[marks[0]+1:marks[-1]]
busy=sum(e-s for s,e,_ in seg); span=seg[-1][1]-seg[0][0]
print("launches per step %.1f  kernel time per step %.1f us  span per step %.1f us  (gaps %.1f us)" % (len(seg)/6, busy/6e3, span/6e3, (span-busy)/6e3))
PY
rm -f $O/run_kernel_trace.csv
cut -d, -f1-5 $O/stats.csv | cut -c1-150
