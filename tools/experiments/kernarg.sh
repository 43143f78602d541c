# launch-latency knobs of the HIP runtime against the pipelined step (64 streams) and the batch-1 latency
for v in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== $v"
  env $v python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['p50_latency_ms'], d['latency_b1']['p50_latency_ms'])"
done
