#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in cfg2 cfg3 cfg5; do
  ST="--steps 8 --warmup 2"; [ $c = cfg3 ] && ST="--steps 3 --warmup 1"
  ( timeout 400 python bench.py --config $c --no-cpu-baseline --no-clock-probe $ST 2>/dev/null | tail -1 ) > gpurun_out/r5c4_bench_$c.json
  python -c "
import json,sys
d=json.load(open('gpurun_out/r5c4_bench_$c.json')); r=d['roofline']
print('$c: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))"
done
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r5c4_tests.txt
cat gpurun_out/r5c4_tests.txt
