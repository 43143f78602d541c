cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run21; rm -rf $O; mkdir -p $O
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 > $O/bench.json
python3 - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3_run21/bench.json'))
print(d['ms_per_step'], d['value'])
for k in d['roofline']['matrix_kernels']:
    print("%-60s n/step %4.1f us %7.1f ms/step %6.3f tflops %6.1f frac %.2f" % (k['kernel'][:60], k['launches_per_step'], k['us_per_launch'], k['ms_per_step'], k['tflops'], k['frac']))
PY
