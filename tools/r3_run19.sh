cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run19; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for i in 1 2; do python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 >> $O/bench.log; done
for i in 1 2; do CONAN_RB_NOLIMB=1 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 >> $O/bench_nolimb.log; done
tail -5 $O/pytest.log; python3 - <<'PY'
import json
for f in ("bench.log","bench_nolimb.log"):
    for l in open("/root/repo/gpurun_out/r3_run19/"+f):
        try: d=json.loads(l); print(f, d["ms_per_step"], d["value"], d.get("p50_latency_ms"), d["roofline"]["kernel"] if "kernel" in d["roofline"] else "")
        except Exception as e: print(f, "bad", l[:200])
PY
