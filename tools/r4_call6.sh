#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gemm_b16_gpu.py tests/test_gemm_x3p_gpu.py tests/test_b16_ops_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -5 gpurun_out/r4/t_b16_ops.log
timeout 300 python tools/bench_gemm_b16.py > gpurun_out/r4/gemm_b16_touch.txt 2>&1; cat gpurun_out/r4/gemm_b16_touch.txt
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_b16_v4.json 2> gpurun_out/r4/bench_cfg5_b16_v4.err
ROOT=$(pwd)
PMC_DRIVER=b16 bash tools/pmc_gemm.sh $ROOT/gpurun_out/r4/pmc_b16_touch.txt
