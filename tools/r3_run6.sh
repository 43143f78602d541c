cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run6; rm -rf $O; mkdir -p $O
run() { python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 6 2>> $O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['ms_per_step'],4), round(d['p50_latency_ms'],3))"; }
run base
CONAN_RC_WIDE_MIN=256 run wide256
CONAN_RC_WIDE_MIN=512 run wide512
CONAN_EMF_CLUSTER=1 run emf1
CONAN_EMF_CLUSTER=2 run emf2
CONAN_RC_WIDE_MIN=256 CONAN_EMF_CLUSTER=2 run wide256_emf2
