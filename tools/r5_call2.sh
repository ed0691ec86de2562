#!/bin/bash
# round 5, GPU call 2: phases of the wide x3 kernel, cfg2 / cfg3 A/B with the slab counts chosen for it, full GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/gemm_x3w_phases.py 2>&1 | tail -20 ) > gpurun_out/r5c2_phases.txt
( BENCH_ARGS="--no-clock-probe" timeout 500 tools/ab_env.sh 2 "PULSE_X3_WIDE=0" "PULSE_X3_WIDE=1" 2>&1 | tail -12 ) > gpurun_out/r5c2_cfg2_ab.txt
( BENCH_ARGS="--no-clock-probe --config cfg3 --steps 3 --warmup 1" timeout 500 tools/ab_env.sh 1 "PULSE_X3_WIDE=0" "PULSE_X3_WIDE=1" 2>&1 | tail -12 ) > gpurun_out/r5c2_cfg3_ab.txt
cat gpurun_out/r5c2_phases.txt gpurun_out/r5c2_cfg2_ab.txt gpurun_out/r5c2_cfg3_ab.txt
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r5c2_tests.txt
cat gpurun_out/r5c2_tests.txt
