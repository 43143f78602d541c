# What each front-end stage costs the pipelined step: bench.py with CONAN_SKIP_STAGE 0 / 1 (no Emformer) / 2 (no decoder) / 3 (vocoder only).
# Timing only - the skipped stages leave their outputs unwritten.  Needs a developer build of the library (make -C conan_amd/csrc clean all DEV=1:
# the shipped library ignores CONAN_SKIP_STAGE).  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd /root/repo
for sk in 0 1 2 3; do
CONAN_SKIP_STAGE=$sk python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 --no-other 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skip $sk', d['ms_per_step'], d.get('p50_latency_ms'))"
done
