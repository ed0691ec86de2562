#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_gemm_b16_gpu.py tests/test_gemm_x3p_gpu.py tests/test_bf16_gpu.py tests/test_network_z_gpu.py tests/test_pulse_kin_gpu.py tests/test_agent_parity_gpu.py tests/test_agent_parity2_gpu.py tests/test_terrain_gpu.py tests/test_task_envs_gpu.py tests/test_env_kernels_gpu.py -x -q > gpurun_out/r4/t_epi.log 2>&1; tail -4 gpurun_out/r4/t_epi.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in d['roofline']['by_variant'].items()})"
done
