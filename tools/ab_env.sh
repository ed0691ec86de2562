#!/bin/bash
# A/B timing of environment-variable settings inside ONE gpurun call (alternating):   tools/ab_env.sh 2 "PULSE_L1_PLANAR=0" "PULSE_L1_PLANAR=1"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    env $v python $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v round $r: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s  avg %.1f us  clock %.3f GHz  kernel share %.3f' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us'], r.get('sustained_clock_ghz') or 0, r['kernel_time_frac_of_step']))"
  done
done
