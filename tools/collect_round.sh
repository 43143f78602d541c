# Round-end evidence on a MI355X box (run from the repo root through gpurun): GPU test log, the default bench line, one
# bench line per BASELINE workload, then the rocprofv3 kernel stats + PMC passes of the default workload
# (tools/collect_profiles.sh).  Everything lands in gpurun_out/profiles_<round>/ with the names profiles/ uses.
R=${R:-r3}; export ROUND=$R
O=gpurun_out/profiles_$R; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rs 2>&1 | grep -v "amdgpu.ids" | tail -8 > $O/${R}_pytest_gpu.txt
# the same suite with the bf16-limb kernels switched on (every parity test and golden at its unchanged tolerance)
CONAN_RB_LIMB=1 timeout 1200 python -m pytest tests -m gpu -q -rs 2>&1 | grep -v "amdgpu.ids" | tail -8 > $O/${R}_pytest_gpu_bf16x3.txt
bash tools/collect_profiles.sh b64 > $O/collect_b64.log 2>&1       # first: the bench lines below read profiles/${R}_b64_pmc.json
bash tools/collect_profiles.sh b64_bf16x3 > $O/collect_b64_bf16x3.log 2>&1
timeout 900 python bench.py 2> $O/bench_default.err | grep '^{' > $O/${R}_b64_bench.json
for W in b64_bf16x3 b1 b1win b128s2 b128s2win b128s2mem4; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline 2> $O/bench_$W.err | grep '^{' > $O/${R}_${W}_bench.json
done
cat $O/${R}_pytest_gpu.txt $O/${R}_pytest_gpu_bf16x3.txt; cut -c1-300 $O/${R}_b64_bench.json; tail -30 $O/collect_b64.log
