#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemm_x3_wide_gpu.py tests/test_learner_kernels_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r5c3_tests.txt
cat gpurun_out/r5c3_tests.txt
( timeout 300 python tools/gemm_x3w_phases.py 2>&1 | tail -20 ) > gpurun_out/r5c3_phases.txt
( timeout 400 python tools/bench_gemm_x3_wide.py --iters 15 2>&1 | tail -60 ) > gpurun_out/r5c3_gemm_ab.txt
( BENCH_ARGS="--no-clock-probe" timeout 500 tools/ab_env.sh 2 "PULSE_X3_WIDE=0" "PULSE_X3_WIDE=1" 2>&1 | tail -12 ) > gpurun_out/r5c3_cfg2_ab.txt
cat gpurun_out/r5c3_phases.txt gpurun_out/r5c3_gemm_ab.txt gpurun_out/r5c3_cfg2_ab.txt
