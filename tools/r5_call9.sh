#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_agent_parity_gpu.py tests/test_agent_parity2_gpu.py tests/test_amp_agent_gpu.py tests/test_bf16_gpu.py tests/test_network_z_gpu.py tests/test_runner_gpu.py -x -q 2>&1 | tail -5 ) > gpurun_out/r5c9_tests.txt
cat gpurun_out/r5c9_tests.txt
for c in cfg2 cfg5 cfg3; do
  ST="--steps 8 --warmup 3"; [ $c = cfg3 ] && ST="--steps 3 --warmup 2"
  for v in 0 1 0 1; do
    PULSE_PLAN_GRAPHS=$v timeout 400 python bench.py --config $c --no-cpu-baseline --no-clock-probe $ST 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$c graphs=$v: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))"
  done
done 2>&1 | tee gpurun_out/r5c9_graphs_ab.txt
