cd /tmp && export TMPDIR=/tmp
cd /root/repo
for v in "CONAN_EMF_CLUSTER=1" "CONAN_EMF_CLUSTER=2" "CONAN_EMF_CLUSTER=4" "CONAN_MEGA_GRID=64" "CONAN_MEGA_GRID=256 CONAN_MEGA_GS=16" "CONAN_MEGA_GS=4" "CONAN_DEC_MEGA=0"; do
env $v python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('p50_latency_ms'))"
done
