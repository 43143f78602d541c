cd /tmp && export TMPDIR=/tmp
cd /root/repo
for v in "X=1" "CONAN_RB_NOLIMB=1"; do
env $v python3 bench.py --workload b1 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('p50_latency_ms'), d.get('latency_stats'))
for k in d['roofline']['matrix_kernels'][:14]: print('   %-55s n %4.1f us %7.1f' % (k['kernel'][:55], k['launches_per_step'], k['us_per_launch']))"
done
