cd /root/repo
for r in 0 1; do for b in cl_bench_first cl_bench_peel cl_bench_first cl_bench_peel; do
  for d in 1 5; do echo "== $b dil $d resid $r"; tools/bin/$b 64 $d 30 0 $r 2>&1 | tail -1; done
done; done
