#!/bin/bash
# round-6 profile evidence (run ON THE GPU BOX): kernel tables + PMC traffic (the two x3 tilings separated) of cfg2 / cfg3 / cfg5, env-kernel
# counters, phases of the wide kernel, per-shape GEMM tables, isolated kernel rooflines, the skinny-N kernel's table
mkdir -p gpurun_out/r6p
ROOT=$(pwd)
bash tools/profile_round.sh r06 cfg2 > gpurun_out/r6p/profile_cfg2.log 2>&1
bash tools/profile_round.sh r06 cfg3 > gpurun_out/r6p/profile_cfg3.log 2>&1
bash tools/profile_round.sh r06 cfg5 > gpurun_out/r6p/profile_cfg5.log 2>&1
PULSE_DISC_STREAM=0 SKIP_PMC=1 bash tools/profile_round.sh r06onechain cfg5 > gpurun_out/r6p/profile_cfg5_onechain.log 2>&1
PMC_KERNEL_FILTER="im_step traj_step amp_obs motion_state" bash tools/pmc_env.sh $ROOT/gpurun_out/r06_env_pmc_counters.txt > gpurun_out/r6p/pmc_env.log 2>&1
timeout 300 python tools/gemm_x3w_phases.py > gpurun_out/r06_gemm_x3w_phases.txt 2> gpurun_out/r6p/phases.err
timeout 300 python tools/gemm_shapes_ab.py cfg2 10 2>/dev/null | grep -v amdgpu > gpurun_out/r06_gemm_shapes_cfg2.txt
timeout 300 python tools/gemm_shapes_ab.py cfg3 6 2>/dev/null | grep -v amdgpu > gpurun_out/r06_gemm_shapes_cfg3.txt
timeout 300 python tools/bench_kernels.py > gpurun_out/r06_kernel_roofline.md 2> gpurun_out/r6p/bench_kernels.err
timeout 200 python tools/bench_gemm_x3_skinny.py --exp 2>/dev/null | grep -v amdgpu > gpurun_out/r06_gemm_x3_skinny.txt
ls -la gpurun_out | grep r06
