for r in 1 2; do for n in A B; do
  echo "== $n"
  ./tools/gemm_bench_exp$n --x3 --relu --clocks --only "x3   upd  fwdL1" 2>&1 | grep -v "census\|options"
  ./tools/gemm_bench_exp$n --x3 --relu --only "x3   upd" 2>&1 | grep -v "census\|options" | tail -1
done; done
