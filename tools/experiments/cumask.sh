for cfg in "0 0" "2 0" "4 0" "8 0" "4 4" "2 2" "0 4"; do
  set -- $cfg
  echo "== front stride $1 emf stride $2"
  CONAN_FRONT_CUSTRIDE=$1 CONAN_EMF_CUSTRIDE=$2 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-b1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value'])"
done
