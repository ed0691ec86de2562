#!/bin/bash
# round 5, GPU call 1: correctness of the 256 x 256 x3 tile, per-shape A/B, cfg2 A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemm_x3_wide_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r5c1_tests.txt
( timeout 400 python tools/bench_gemm_x3_wide.py --iters 15 2>&1 | tail -60 ) > gpurun_out/r5c1_gemm_ab.txt
( BENCH_ARGS="--no-clock-probe" timeout 500 tools/ab_env.sh 2 "PULSE_X3_WIDE=0" "PULSE_X3_WIDE=1" 2>&1 | tail -12 ) > gpurun_out/r5c1_cfg2_ab.txt
cat gpurun_out/r5c1_tests.txt gpurun_out/r5c1_gemm_ab.txt gpurun_out/r5c1_cfg2_ab.txt
