cd /tmp && export TMPDIR=/tmp
cd /root/repo
cp conan_amd/libconan_hip.so /tmp/orig.so
for v in orig s8 s32 orig; do
if [ $v = orig ]; then cp /tmp/orig.so conan_amd/libconan_hip.so; else cp conan_amd/libconan_hip_$v.so conan_amd/libconan_hip.so; fi
python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('p50_latency_ms'))"
done
cp /tmp/orig.so conan_amd/libconan_hip.so
