cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run22; rm -rf $O; mkdir -p $O
for sh in "" 0 1 2; do
CONAN_CL_SHAPE=$sh python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 > $O/bench$sh.json
python3 - "$sh" <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r3_run22/bench%s.json' % sys.argv[1]))
print("shape", sys.argv[1] or "auto", d['ms_per_step'], d['value'])
for k in d['roofline']['matrix_kernels']:
    if 'conv_' in k['kernel']: print("   %-50s n/step %4.1f us %7.1f ms/step %6.3f tflops %6.1f" % (k['kernel'][:50], k['launches_per_step'], k['us_per_launch'], k['ms_per_step'], k['tflops']))
PY
done
python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_configs.py tests/test_gpu.py -x -q 2>&1 | tail -2
