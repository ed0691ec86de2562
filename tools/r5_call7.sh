#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_x3_wide_gpu.py tests/test_learner_kernels_gpu.py tests/test_network_z_gpu.py tests/test_pulse_kin_gpu.py tests/test_vae_head_gpu.py "tests/test_full_size_parity_gpu.py" -x -q 2>&1 | tail -6 ) > gpurun_out/r5c7_tests.txt
cat gpurun_out/r5c7_tests.txt
for v in 0 1 0 1; do
  PULSE_SILU_DERIV=$v timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-clock-probe --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg3 siluderiv=$v: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))"
done 2>&1 | tee gpurun_out/r5c7_cfg3_ab.txt
python tools/gemm_shapes_ab.py cfg3 6 2>&1 | tail -48 > gpurun_out/r5c7_shapes_cfg3.txt; head -24 gpurun_out/r5c7_shapes_cfg3.txt
