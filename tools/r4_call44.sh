#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_b16_ops_gpu.py tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py tests/test_disc_gpu.py tests/test_agent_parity2_gpu.py tests/test_full_size_parity_gpu.py -q > gpurun_out/r4/t_c44.log 2>&1; tail -5 gpurun_out/r4/t_c44.log
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r04_bench_cfg5.json 2> gpurun_out/r4/bench_cfg5_final.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_cfg5.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5', round(d['ms_per_step'],2), round(d['value']), d['per_step_play_update_ms'])"
done
