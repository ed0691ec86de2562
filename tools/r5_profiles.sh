#!/bin/bash
# round-5 profile evidence (run ON THE GPU BOX): kernel tables + PMC traffic of cfg2 / cfg3 / cfg5, counters of both x3 tilings, phases of the wide kernel
mkdir -p gpurun_out/r5p
ROOT=$(pwd)
bash tools/profile_round.sh r05 cfg2 > gpurun_out/r5p/profile_cfg2.log 2>&1
bash tools/profile_round.sh r05 cfg3 > gpurun_out/r5p/profile_cfg3.log 2>&1
bash tools/profile_round.sh r05 cfg5 > gpurun_out/r5p/profile_cfg5.log 2>&1
PULSE_DISC_STREAM=0 SKIP_PMC=1 bash tools/profile_round.sh r05onechain cfg5 > gpurun_out/r5p/profile_cfg5_onechain.log 2>&1
PMC_DRIVER=x3w bash tools/pmc_gemm.sh $ROOT/gpurun_out/r05_gemm_x3_tilings_pmc_counters.txt > gpurun_out/r5p/pmc_x3w.log 2>&1
PMC_KERNEL_FILTER="im_step traj_step amp_obs motion_state" bash tools/pmc_env.sh $ROOT/gpurun_out/r05_env_pmc_counters.txt > gpurun_out/r5p/pmc_env.log 2>&1
timeout 300 python tools/gemm_x3w_phases.py > gpurun_out/r05_gemm_x3w_phases.txt 2> gpurun_out/r5p/phases.err
timeout 400 python tools/bench_gemm_x3_wide.py --iters 15 > gpurun_out/r05_gemm_x3_wide_ab.txt 2> gpurun_out/r5p/ab.err
timeout 300 python tools/gemm_shapes_ab.py cfg2 10 2>/dev/null | grep -v amdgpu > gpurun_out/r05_gemm_shapes_cfg2.txt
timeout 300 python tools/gemm_shapes_ab.py cfg3 6 2>/dev/null | grep -v amdgpu > gpurun_out/r05_gemm_shapes_cfg3.txt
ls -la gpurun_out | grep r05
