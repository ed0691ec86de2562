#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gemm_b16_gpu.py -x -q > gpurun_out/r4/t_c27.log 2>&1; tail -5 gpurun_out/r4/t_c27.log
for t in 1 2; do echo "--- tile option $t"; timeout 300 python tools/bench_gemm_b16.py --tile $t 2>&1 | grep -v amdgpu | cut -c1-120; done
