#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_agent_parity2_gpu.py tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py tests/test_disc_gpu.py tests/test_b16_ops_gpu.py tests/test_full_size_parity_gpu.py -q > gpurun_out/r4/t_c30.log 2>&1; tail -8 gpurun_out/r4/t_c30.log
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in r['by_variant'].items()})"
