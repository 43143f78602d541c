#!/bin/bash
# Same-box A/B of developer plans on the headline workload, alternating runs:  bash tools/ab_plan.sh "" "MEGA_LAYOUT=g" [...]
# (each argument is a conan_streams_opts.dev_plan string; "" = the default plan).  N=3 rounds by default.
N=${N:-3}
P="import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-28s ms/step %.4f  p50 %.3f  vocoder alone %.3f  front-end cost %.3f' % (repr(sys.argv[1]), d['ms_per_step'], d['p50_latency_ms'], r['vocoder_alone_ms'], r['frontend_cost_ms']))"
for i in $(seq $N); do
  for plan in "$@"; do
    if [ -z "$plan" ]; then A=""; else A="--dev-plan $plan"; fi
    timeout 600 python bench.py --no-cpu-baseline --no-b1 --no-other --steps 60 --warmup 10 $A $AB_FLAGS 2>/dev/null | grep '^{' | python3 -c "$P" "$plan"
  done
done
