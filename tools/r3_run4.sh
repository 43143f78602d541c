cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run4; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_round3.py -x -q -k "wide_stage" 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 10 > $O/bench_pair$i.json 2> $O/bench.err
CONAN_RB_NOPAIR=1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 10 > $O/bench_nopair$i.json 2>> $O/bench.err
done
python3 -c "
import json
for f in ('pair1','nopair1','pair2','nopair2'):
    d=json.loads(open('$O/bench_%s.json'%f).read().strip().splitlines()[-1]);print(f, round(d['ms_per_step'],4),round(d['p50_latency_ms'],3),d['step_time_stats']['p50_ms'])"
