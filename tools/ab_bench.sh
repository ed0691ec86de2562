#!/bin/bash
# A/B timing of library variants inside ONE gpurun call, alternating so box-to-box and warm-up drift cancel:
#   tools/ab_bench.sh 2 s136 s132      (2 rounds; variants are build/variants/libpulse_gemm_<name>.so)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    PULSE_HIP_LIB=$ROOT/build/variants/libpulse_gemm_$v.so python $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v round $r: %.4f M env-steps/s  %.2f ms  GEMM %.1f TF/s  avg %.1f us  clock %.3f GHz  frac@clock %.3f' % (d['value']/1e6, d['ms_per_step'], r['achieved'], r['avg_us'], r.get('sustained_clock_ghz') or 0, r.get('frac_at_sustained_clock') or 0))"
  done
done
