cd /tmp && export TMPDIR=/tmp
cd /root/repo
export CONAN_RB_NOPAIR=1
for sh in 0 3; do
CONAN_CL_SHAPE=$sh python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shape $sh', round(d['ms_per_step'],4), round(d.get('p50_latency_ms'),4))
for k in d['roofline']['matrix_kernels']:
    if 'conv_' in k['kernel']: print('   %-50s n/step %4.1f us %7.1f ms/step %6.3f' % (k['kernel'][:50], k['launches_per_step'], k['us_per_launch'], k['ms_per_step']))"
done
