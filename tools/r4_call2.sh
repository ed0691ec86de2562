#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_b16_ops_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -15 gpurun_out/r4/t_b16_ops.log
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_disc_gpu.py tests/test_amp_agent_gpu.py -q > gpurun_out/r4/t_bf16.log 2>&1; tail -15 gpurun_out/r4/t_bf16.log
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_b16.json 2> gpurun_out/r4/bench_cfg5_b16.err; tail -3 gpurun_out/r4/bench_cfg5_b16.err; cut -c1-600 gpurun_out/r4/bench_cfg5_b16.json
