cd /tmp && export TMPDIR=/tmp
cd /root/repo
O=$PWD/gpurun_out/r3_run5; rm -rf $O; mkdir -p $O
for sk in 0 1 2 3; do for np in 0 1; do
  if [ $np = 1 ]; then export CONAN_RB_NOPAIR=1; else unset CONAN_RB_NOPAIR; fi
  CONAN_SKIP_STAGE=$sk python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-b1 --latency-steps 2 > $O/b_${sk}_${np}.json 2>> $O/bench.err
done; done
python3 -c "
import json
for sk in range(4):
  for np in range(2):
    d=json.loads(open('$O/b_%d_%d.json'%(sk,np)).read().strip().splitlines()[-1]);print('skip',sk,'nopair',np, round(d['ms_per_step'],4))"
