#!/bin/bash
# round-4 profile evidence (run ON THE GPU BOX): kernel tables + PMC traffic of cfg2 / cfg5, counters of the bf16-storage GEMM and the env step
mkdir -p gpurun_out/r4p
ROOT=$(pwd)
bash tools/profile_round.sh r04 cfg2 > gpurun_out/r4p/profile_cfg2.log 2>&1
bash tools/profile_round.sh r04 cfg5 > gpurun_out/r4p/profile_cfg5.log 2>&1
# the same command with the discriminator chain inline (one chain at a time: the kernels' own durations)
PULSE_DISC_STREAM=0 SKIP_PMC=1 bash tools/profile_round.sh r04onechain cfg5 > gpurun_out/r4p/profile_cfg5_onechain.log 2>&1
# steady-state dispatches per epoch: the 3-epoch traces above include the one-time demo-buffer fill (391 fetches) and first-use setup; a 6-epoch
# trace of the same command minus the 3-epoch one leaves three steady epochs
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_cfg5/trace6 -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 5 --warmup 1 --no-clock-probe > /dev/null 2> $ROOT/gpurun_out/r4p/trace6.err
  DB=$(find $ROOT/gpurun_out/prof_cfg5/trace6 -name "*.db" | head -1); python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4p/kstats_cfg5_6epochs.md > /dev/null; rm -rf $ROOT/gpurun_out/prof_cfg5/trace6 )
python - <<'PY'
import re
def disp(p):
    t = open(p).read()
    m = re.search(r"(\d+) dispatches", t)
    return int(m.group(1))
a, b = disp("gpurun_out/r04_bench_kernel_stats_cfg5.md"), disp("gpurun_out/r4p/kstats_cfg5_6epochs.md")
open("gpurun_out/r04_dispatches_cfg5.txt", "w").write(
    f"# cfg5 dispatches (rocprofv3 --kernel-trace): {a} over warm-up + 2 epochs (includes the one-time AMP demo-buffer fill: 391 fetches x ~17 launches),\n"
    f"# {b} over warm-up + 5 epochs -> steady state {(b - a) / 3:.0f} dispatches per epoch\n")
print(open("gpurun_out/r04_dispatches_cfg5.txt").read())
PY
PMC_DRIVER=b16 bash tools/pmc_gemm.sh $ROOT/gpurun_out/r04_gemm_b16_pmc_ring.txt > gpurun_out/r4p/pmc_b16.log 2>&1
bash tools/pmc_env.sh $ROOT/gpurun_out/r04_env_pmc_counters.txt > gpurun_out/r4p/pmc_env.log 2>&1
timeout 300 python tools/bench_kernels.py > gpurun_out/r04_kernel_roofline.md 2> gpurun_out/r4p/bench_kernels.err
ls -la gpurun_out | grep r04_
