# pipelined step time at 64 streams against the Emformer cluster size (CONAN_EMF_CLUSTER: workgroups per stream pair)
for c in 0 1 2 4 8; do
  echo "== CONAN_EMF_CLUSTER=$c"
  CONAN_EMF_CLUSTER=$c python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-b1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['p50_latency_ms']);
[print(k['kernel'],k['launches_per_step'],round(k['us_per_launch'],1)) for k in d['roofline']['matrix_kernels'] if 'emformer' in k['kernel']]"
done
