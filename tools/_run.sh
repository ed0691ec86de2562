mkdir -p gpurun_out
for n in 1 2 3 4 8 15; do
  echo "== exp $n" 
  ./tools/gemm_bench_exp$n --x3 --clocks --only "x3   upd  fwdL1" 2>&1 | grep -v census
  ./tools/gemm_bench_exp$n --x3 --clocks --only "x3   big" 2>&1 | grep -v census
done
