cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run2; rm -rf $O; mkdir -p $O
echo "== rb_bench default" > $O/rb.txt; tools/bin/rb_bench_x 64 4 20 2>&1 | grep -v "^   " | head -4 >> $O/rb.txt
echo "== rb_bench C=128 rows=32" >> $O/rb.txt; RB_ROWS=32 RB_ROWS_C=128 tools/bin/rb_bench_x 64 4 20 2>&1 | head -9 >> $O/rb.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4 > $O/pytest.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 tools/blocking_trace.py 64 > $O/trace.log 2>&1
python3 tools/trace_seq.py $O/trace/run_kernel_trace.csv "conv_mfma_kernel<32, 64" "conv_mfma_kernel<64, 64, 2, 2, 1, 32" > $O/seq_ups.txt 2>&1
rm -f $O/trace/run_kernel_trace.csv
python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/rb.txt $O/pytest.txt $O/seq_ups.txt
python3 -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['p50_latency_ms'],d['step_time_stats'],d['latency_b1']['p50_latency_ms'])
for k in d['roofline']['matrix_kernels']: print(k['kernel'],k['launches_per_step'],round(k['us_per_launch'],1),round(k['frac'],3))"
