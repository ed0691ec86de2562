#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gemm_b16_gpu.py tests/test_gemm_x3p_gpu.py tests/test_b16_ops_gpu.py tests/test_bf16_gpu.py tests/test_disc_gpu.py tests/test_amp_agent_gpu.py tests/test_learner_kernels_gpu.py tests/test_agent_parity_gpu.py -x -q > gpurun_out/r4/t_c14.log 2>&1; tail -4 gpurun_out/r4/t_c14.log
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(d['roofline']['achieved'],1))"
done
timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(d['roofline']['achieved'],1))"
