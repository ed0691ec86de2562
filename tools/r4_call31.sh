#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_agent_parity_gpu.py tests/test_runner_gpu.py tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py -q -x > gpurun_out/r4/t_c31.log 2>&1; tail -4 gpurun_out/r4/t_c31.log
for w in 0 1 0 1; do
PULSE_BWD_STREAM=$w timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 bwdstream=$w', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1))"
done
for w in 0 1 0 1; do
PULSE_BWD_STREAM=$w timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5 bwdstream=$w', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1))"
done
