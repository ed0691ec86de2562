#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gemm_b16_gpu.py -x -q > gpurun_out/r4/t_c26.log 2>&1; tail -5 gpurun_out/r4/t_c26.log
for t in 1 2; do echo "--- tile option $t"; timeout 300 python tools/bench_gemm_b16.py --tile $t 2>&1 | grep -v amdgpu | cut -c1-120 | awk '{print $1,$2,$3,$4, $(NF-2), $(NF-1)}'; done
for w in 0 1 0 1; do
PULSE_B16_WIDE=$w timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5 wide=$w', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in r['by_variant'].items()})"
done
