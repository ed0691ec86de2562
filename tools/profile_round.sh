#!/bin/bash
# One round of rocprofv3 evidence for one bench config, run ON THE GPU BOX (gpurun):
#   tools/profile_round.sh r02 cfg2
# writes gpurun_out/<round>_bench_kernel_stats_<cfg>.md  (kernel trace, --stats)
#        gpurun_out/<round>_gemm_traffic_<cfg>.json      (PMC: FETCH_SIZE / WRITE_SIZE in separate passes, + TCC hit / miss pass)
#        gpurun_out/<round>_bench_<cfg>_profiled.json    (the bench line printed under the kernel trace)
# Counter passes use --pmc with --kernel-trace only (never mixed with other trace domains).
set -u
R=$1; CFG=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof_$CFG
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --config $CFG --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe"
rocprofv3 --kernel-trace --stats -d $OUT/prof_$CFG/trace -- $BENCH > $OUT/${R}_bench_${CFG}_profiled.json 2> $OUT/prof_$CFG/trace.err
DB=$(find $OUT/prof_$CFG/trace -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $CFG --no-cpu-baseline --steps 2 --warmup 1   ($CFG, MI355X, $R)" > $OUT/${R}_bench_kernel_stats_${CFG}.md
echo >> $OUT/${R}_bench_kernel_stats_${CFG}.md
python $ROOT/tools/rocprof_summary.py "$DB" $OUT/${R}_bench_kernel_stats_${CFG}.md > /dev/null
[ "${SKIP_PMC:-0}" = "1" ] && { rm -rf $OUT/prof_$CFG/trace; ls -la $OUT | grep ${R}_; exit 0; }      # kernel table only
PW=1; [ "$CFG" != "cfg2" ] && PW=0    # the 8192-env configs crashed rocprofv3 itself in counter mode with a warm-up epoch: profile a single epoch
PMC="python $ROOT/bench.py --config $CFG --no-cpu-baseline --steps 1 --warmup $PW --no-roofline"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -d $OUT/prof_$CFG/pmc_$T --output-format csv -- $PMC > /dev/null 2> $OUT/prof_$CFG/pmc_$T.err
done
python $ROOT/tools/pmc_traffic.py $OUT/prof_$CFG $OUT/${R}_gemm_traffic_${CFG}.json > /dev/null
for T in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do tail -c 2000 $OUT/prof_$CFG/pmc_$T.err > $OUT/prof_$CFG/err_$T.txt 2>/dev/null; done
rm -rf $OUT/prof_$CFG/trace $OUT/prof_$CFG/pmc_*   # raw traces are large: only the summaries travel back
ls -la $OUT | grep ${R}_
