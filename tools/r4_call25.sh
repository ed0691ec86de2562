#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gemm_b16_gpu.py -x -q > gpurun_out/r4/t_c25.log 2>&1; tail -3 gpurun_out/r4/t_c25.log
timeout 300 python tools/bench_gemm_b16.py 2>&1 | grep -v amdgpu | cut -c1-120
ROOT=$(pwd)
PMC_DRIVER=b16 bash tools/pmc_gemm.sh $ROOT/gpurun_out/r04_gemm_b16_pmc_ring.txt > /dev/null 2>&1
grep -n "^##\|MFMA pipe busy\|SQ_LDS_BANK\|SQ_LDS_IDX\|SQ_WAIT_ANY /" $ROOT/gpurun_out/r04_gemm_b16_pmc_ring.txt
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 --no-clock-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in r['by_variant'].items()})"
done
