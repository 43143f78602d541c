# Round-end evidence on a MI355X box (run from the repo root through gpurun): GPU test log, the default bench line, one
# bench line per BASELINE workload, then the rocprofv3 kernel stats + PMC passes of the default workload in both arithmetic
# forms (tools/collect_profiles.sh).  Everything lands in gpurun_out/profiles_<round>/ with the names profiles/ uses.
R=${R:-r6}; export ROUND=$R
O=gpurun_out/profiles_$R; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | grep -v "amdgpu.ids" | tail -12 > $O/${R}_pytest_gpu.txt
# the float64 error tables of both arithmetic forms (tests/test_gpu_arith.py prints them)
timeout 900 python -m pytest tests/test_gpu_arith.py -q -s 2>&1 | grep -v "amdgpu.ids" > $O/${R}_arith_vs_f64.txt
# (the per-operator stamps of the decoder megakernel - tools/mega_probe.py, tools/xcd_check.py - and the l2 / layout A/Bs of round 5 read
# developer switches from the environment: `make DEV=1` builds only since ABI 8)
# the default configuration at 1 .. 128 streams per GPU
bash tools/stream_sweep.sh > $O/${R}_stream_sweep.txt 2>/dev/null
# what CONAN_STREAMS_FIXED_PLAN costs at 64 of 64 slots active (alternating runs)
for i in 1 2 3; do for F in "" "--fixed-plan"; do
  timeout 600 python bench.py --no-cpu-baseline --no-other --no-b1 $F 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed_plan=%s ms/step %.4f p50 %.3f vocoder alone %.3f' % (d['config']['fixed_plan'], d['ms_per_step'], d['p50_latency_ms'], d['roofline']['vocoder_alone_ms']))"
done; done > $O/${R}_fixed_plan_cost.txt
timeout 300 python tools/stage_times.py 64 2>&1 | grep -v "amdgpu.ids" | grep -v "^  op\|decoder_mega\]" > $O/${R}_stage_times.txt
bash tools/collect_profiles.sh b64 auto > $O/collect_b64.log 2>&1       # first: the bench lines below read profiles/${R}_b64_pmc.json
bash tools/collect_profiles.sh b64 f32 > $O/collect_b64_f32.log 2>&1
timeout 900 python bench.py 2> $O/bench_default.err | grep '^{' > $O/${R}_b64_bench.json
timeout 900 python bench.py --arith f32 --no-cpu-baseline --no-other 2> $O/bench_f32.err | grep '^{' > $O/${R}_b64_f32_bench.json
CONAN_BENCH_COMM=1 timeout 900 python bench.py --no-cpu-baseline --no-other --no-b1 2> $O/bench_comm.err | grep '^{' > $O/${R}_b64_bench_comm.json
for W in b1 b1win b128s2 b128s2win b128s2mem4; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline 2> $O/bench_$W.err | grep '^{' > $O/${R}_${W}_bench.json
done
# the driver's own parameters (K = 20, W = 5), and one blocking one-stream step as a dispatch timeline
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other 2> $O/bench_k20.err | grep '^{' > $O/${R}_b64_bench_k20.json
( cd /tmp && export TMPDIR=/tmp; cd /root/repo; rm -rf gpurun_out/${R}_tl1; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_tl1 -o run -- python3 tools/blocking_trace.py 1 > gpurun_out/${R}_tl1.log 2>&1; python3 tools/trace_timeline.py gpurun_out/${R}_tl1/run_kernel_trace.csv 6 > $O/${R}_b1_step_timeline.txt; rm -rf gpurun_out/${R}_tl1 )
cat $O/${R}_pytest_gpu.txt; cut -c1-300 $O/${R}_b64_bench.json; tail -30 $O/collect_b64.log
