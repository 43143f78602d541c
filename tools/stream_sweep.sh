#!/bin/bash
# The default configuration at 1 .. 128 streams per GPU (pipelined ms per step, blocking p50, vocoder alone, front-end cost):
#   bash tools/stream_sweep.sh [sizes ...] > profiles/rN_stream_sweep.txt       (extra bench.py flags: SWEEP_FLAGS="--dev-plan ...")
SIZES=${@:-1 2 3 4 8 15 16 17 24 32 40 48 64 96 128}
for n in $SIZES; do
  timeout 600 python bench.py --streams $n --no-cpu-baseline --no-other --no-b1 $SWEEP_FLAGS 2>/dev/null | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('streams %3d  %s ms/step %.4f  chunks/s %8.0f  blocking p50 %.3f  vocoder alone %.3f  front-end cost %.3f' % (d['config']['streams_per_gpu'], d['config']['arith'], d['ms_per_step'], d['value'], d['p50_latency_ms'], r['vocoder_alone_ms'], r['frontend_cost_ms']))"
done
