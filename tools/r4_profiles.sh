#!/bin/bash
# round-4 profile evidence (run ON THE GPU BOX): kernel tables + PMC traffic of cfg2 / cfg5, counters of the bf16-storage GEMM and the env step
mkdir -p gpurun_out/r4p
ROOT=$(pwd)
bash tools/profile_round.sh r04 cfg2 > gpurun_out/r4p/profile_cfg2.log 2>&1
bash tools/profile_round.sh r04 cfg5 > gpurun_out/r4p/profile_cfg5.log 2>&1
# the same command with the discriminator chain inline (one chain at a time: the kernels' own durations)
PULSE_DISC_STREAM=0 SKIP_PMC=1 bash tools/profile_round.sh r04onechain cfg5 > gpurun_out/r4p/profile_cfg5_onechain.log 2>&1
PMC_DRIVER=b16 bash tools/pmc_gemm.sh $ROOT/gpurun_out/r04_gemm_b16_pmc_ring.txt > gpurun_out/r4p/pmc_b16.log 2>&1
bash tools/pmc_env.sh $ROOT/gpurun_out/r04_env_pmc_counters.txt > gpurun_out/r4p/pmc_env.log 2>&1
timeout 300 python tools/bench_kernels.py > gpurun_out/r04_kernel_roofline.md 2> gpurun_out/r4p/bench_kernels.err
ls -la gpurun_out | grep r04_
