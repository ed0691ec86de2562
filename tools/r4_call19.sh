#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_env_kernels_gpu.py tests/test_env_variants_gpu.py tests/test_motion_env_gpu.py tests/test_getup_gpu.py tests/test_contention_gpu.py tests/test_task_envs_gpu.py tests/test_agent_parity_gpu.py -x -q > gpurun_out/r4/t_c19.log 2>&1; tail -4 gpurun_out/r4/t_c19.log
for v in 0 1 0 1; do echo "PULSE_IM_TWO_ROLES=$v"; PULSE_IM_TWO_ROLES=$v timeout 300 python tools/bench_kernels.py 2>/dev/null | grep -a "im_step"; done
