#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemm_b16_gpu.py tests/test_b16_ops_gpu.py tests/test_gemm_x3p_gpu.py -x -q 2>&1 | tail -4 ) > gpurun_out/r5c6_tests.txt
cat gpurun_out/r5c6_tests.txt
( timeout 300 python tools/gemm_b16_phases.py 2>&1 | tail -12 ) > gpurun_out/r5c6_b16_phases.txt
cat gpurun_out/r5c6_b16_phases.txt
for v in 0 1 0 1; do
  PULSE_BWD_STREAM=$v timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-clock-probe --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg2 bwdstream=$v: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))"
done 2>&1 | tee gpurun_out/r5c6_bwdstream.txt
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-clock-probe --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg5: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))" | tee gpurun_out/r5c6_cfg5.txt
