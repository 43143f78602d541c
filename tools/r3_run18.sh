cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run18; rm -rf $O; mkdir -p $O
CONAN_MEGA_STAMPS=0 python3 tools/mega_probe.py 64 > $O/probe.log 2>&1
python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_configs.py tests/test_gpu_round2.py -x -q > $O/pytest.log 2>&1
for i in 1 2 3; do python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 >> $O/bench.log; done
python3 bench.py --steps 100 --warmup 20 --workload b128s2mem4 --no-cpu-baseline --no-b1 2>$O/mem.err | tail -1 > $O/bench_mem.log
tail -3 $O/probe.log; tail -3 $O/pytest.log; python3 - <<'PY'
import json
for f in ("bench.log","bench_mem.log"):
    for l in open("/root/repo/gpurun_out/r3_run18/"+f):
        try: d=json.loads(l); print(f, d["ms_per_step"], d["value"], d.get("p50_latency_ms"))
        except Exception as e: print(f, "bad", l[:200])
PY
