cd /root/repo
for cfgs in "-1,-1,-1,-1" "-1,-1,0,0" "-1,0,0,0" "0,-1,-1,-1" "-1,-1,2,2" "1,-1,-1,-1" "-1,-1,-1,7"; do
echo "== CONAN_UPS_CFG=$cfgs"
CONAN_UPS_CFG=$cfgs python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-b1 --latency-steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['p50_latency_ms'],3))
for k in d['roofline']['matrix_kernels']:
    if 'conv_mfma' in k['kernel']: print('  ',k['kernel'],k['launches_per_step'],round(k['us_per_launch'],1),round(k['frac'],3))"
done
