#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_b16_ops_gpu.py tests/test_learner_kernels_gpu.py tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py tests/test_disc_gpu.py tests/test_agent_parity_gpu.py tests/test_agent_parity2_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -4 gpurun_out/r4/t_b16_ops.log
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_b16_v7.json 2> gpurun_out/r4/bench_cfg5_b16_v7.err
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r4/bench_cfg2_v7.json 2> gpurun_out/r4/bench_cfg2_v7.err
