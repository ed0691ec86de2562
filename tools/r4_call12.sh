#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_terrain_gpu.py tests/test_getup_gpu.py tests/test_agent_parity_gpu.py tests/test_agent_parity2_gpu.py tests/test_learner_kernels_gpu.py -x -q -s > gpurun_out/r4/t_c12.log 2>&1; grep -a "terrain\]" gpurun_out/r4/t_c12.log; tail -4 gpurun_out/r4/t_c12.log
for v in 0 1 0 1; do
PULSE_DX_FORWARD=$v timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 dxfwd=$v', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in d['roofline']['by_variant'].items()})"
done
