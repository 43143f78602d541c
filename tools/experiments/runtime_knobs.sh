# HIP / ROCr runtime knobs against the pipelined step (64 streams), its blocking p50 and the batch-1 p50.
# (ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT in the list: with it the host never sees the completion signals - the bench hangs.)
for v in "" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=2000" "GPU_MAX_HW_QUEUES=16"; do
  echo "== $v"
  env $v timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['p50_latency_ms'], d['latency_b1']['p50_latency_ms'])"
done
