cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 tools/blocking_trace.py 64 > $O/trace.log 2>&1
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv > $O/seq_all.txt 2>&1
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv "resblock_pair" > $O/seq_rp.txt 2>&1
rm -f $O/trace/run_kernel_trace.csv
python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 tools/stage_times.py 64 > $O/stage_times.txt 2>&1
head -16 $O/seq_all.txt; cat $O/seq_rp.txt; tail -4 $O/stage_times.txt
python3 -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['p50_latency_ms'],d['step_time_stats']['p50_ms'],d['latency_b1']['p50_latency_ms'])
for k in d['roofline']['matrix_kernels']: print(k['kernel'],k['launches_per_step'],round(k['us_per_launch'],1),round(k['frac'],3))"
