# A/B of two builds of the library on ONE box, alternating: bash tools/ab_bench.sh <alt .so>
cd /tmp && export TMPDIR=/tmp
cd /root/repo
ALT=$1
cp conan_amd/libconan_hip.so /tmp/new.so
for v in new alt new alt new alt; do
if [ $v = new ]; then cp /tmp/new.so conan_amd/libconan_hip.so; else cp $ALT conan_amd/libconan_hip.so; fi
python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-b1 --no-other 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d.get('p50_latency_ms'),4), round(d.get('step_time_stats',{}).get('p95_ms',0),3))"
done
cp /tmp/new.so conan_amd/libconan_hip.so
