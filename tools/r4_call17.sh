#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gemm_b16_gpu.py -x -q > gpurun_out/r4/t_c17.log 2>&1; tail -6 gpurun_out/r4/t_c17.log
for v in 0 1; do echo "PULSE_B16_RING=$v"; PULSE_B16_RING=$v timeout 300 python tools/bench_gemm_b16.py 2>&1 | cut -c1-120; done > gpurun_out/r4/gemm_b16_ring.txt; cat gpurun_out/r4/gemm_b16_ring.txt
