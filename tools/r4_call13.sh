#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_motion_env_gpu.py tests/test_env_kernels_gpu.py tests/test_amp_agent_gpu.py tests/test_terrain_gpu.py tests/test_bf16_gpu.py tests/test_agent_parity2_gpu.py -x -q > gpurun_out/r4/t_c13.log 2>&1; tail -4 gpurun_out/r4/t_c13.log
for v in 0 1 0 1; do
PULSE_AMP_FUSED=$v timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 ampfused=$v', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2))"
done
