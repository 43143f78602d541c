cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run23; rm -rf $O; mkdir -p $O
for v in a b; do
if [ $v = a ]; then export CONAN_RB_NOPAIR=1; else unset CONAN_RB_NOPAIR; fi
CONAN_CL_SHAPE=$1 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 > $O/bench$v.json
python3 - "$v" <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r3_run23/bench%s.json' % sys.argv[1]))
print("variant", sys.argv[1], d['ms_per_step'], d['value'], d.get('p50_latency_ms'))
for k in d['roofline']['matrix_kernels']:
    print("   %-55s n/step %4.1f us %7.1f ms/step %6.3f tflops %6.1f" % (k['kernel'][:55], k['launches_per_step'], k['us_per_launch'], k['ms_per_step'], k['tflops']))
PY
done
