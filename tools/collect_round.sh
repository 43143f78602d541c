# Round-end evidence on a MI355X box (run from the repo root through gpurun): GPU test log, the default bench line, one
# bench line per BASELINE workload, then the rocprofv3 kernel stats + PMC passes of the default workload in both arithmetic
# forms (tools/collect_profiles.sh).  Everything lands in gpurun_out/profiles_<round>/ with the names profiles/ uses.
R=${R:-r5}; export ROUND=$R
O=gpurun_out/profiles_$R; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | grep -v "amdgpu.ids" | tail -12 > $O/${R}_pytest_gpu.txt
# the float64 error tables of both arithmetic forms (tests/test_gpu_arith.py prints them)
timeout 900 python -m pytest tests/test_gpu_arith.py -q -s 2>&1 | grep -v "amdgpu.ids" > $O/${R}_arith_vs_f64.txt
# per-operator stamps of the decoder megakernel alone, the stages alone against the pipelined step
timeout 300 python tools/mega_probe.py 64 2>&1 | grep -v "amdgpu.ids" > $O/${R}_decoder_mega_stamps.txt
# ... and of single-tile steps (xcd mode) at one and four streams, with the separate launches beside them
for n in 1 4; do CONAN_MEGA_STAMPS=1 timeout 300 python tools/xcd_check.py $n 2>&1 | grep -v "amdgpu.ids" >> $O/${R}_decoder_xcd_stamps.txt; done
# the one-launch vocoder step (opt-in) against the launch plans
for n in 1 4; do timeout 300 python tools/chain_check.py $n 4 2>&1 | grep "slots" >> $O/${R}_voc_chain.txt; done
bash tools/ab_round5.sh > $O/${R}_ab_l2_groups.txt 2>&1
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
