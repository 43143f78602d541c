# Per-kernel cost of the pipeline: rocprofv3 kernel traces of pipelined steps (bench.py --marks) and of blocking steps, averages over the
# marked steps side by side (the extra time a kernel takes when the other two stages run beside it).  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/pipeline_cost; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/pipe -o run -- python3 bench.py --steps 30 --warmup 10 --marks > $O/pipe.log 2>&1
python3 tools/marked_stats.py $O/pipe/run_kernel_trace.csv 30 > $O/pipe_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/blk -o run -- python3 tools/blocking_trace.py 64 > $O/blk.log 2>&1
python3 tools/marked_stats.py $O/blk/run_kernel_trace.csv 6 > $O/blk_stats.csv
rm -f $O/*/run_kernel_trace.csv
python3 - <<'PY'
import csv
O="/root/repo/gpurun_out/pipeline_cost"
def rd(f): return {r["Name"]:r for r in csv.DictReader(open(f))}
p,b=rd(O+"/pipe_stats.csv"),rd(O+"/blk_stats.csv")
tot=0
for n,r in sorted(p.items(), key=lambda kv:-int(kv[1]["TotalDurationNs"])):
    if n in b and float(r["CallsPerStep"])>=0.9:
        d=(float(r["AverageNs"])-float(b[n]["AverageNs"]))*float(b[n]["CallsPerStep"])/1e3
        tot+=d
        print("%-70s n/step %4.1f  pipelined avg %7.1f us  blocking avg %7.1f us  extra per step %6.1f us  (max %7.1f)" % (n[:70], float(r["CallsPerStep"]), float(r["AverageNs"])/1e3, float(b[n]["AverageNs"])/1e3, d, float(r["MaxNs"])/1e3))
print("sum of extra per step %.1f us" % tot)
PY
