#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_agent_parity_gpu.py tests/test_agent_parity2_gpu.py tests/test_amp_agent_gpu.py tests/test_abi.py -x -q 2>&1 | tail -4 ) > gpurun_out/r5c8_tests.txt
cat gpurun_out/r5c8_tests.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5c8_cfg2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c8_cfg2.json')); r=d['roofline']
print('cfg2 %.2f ms %.4f M | dominant %s: %.1f TF/s frac %.3f (%d launches, %.1f us) | all x3: %s | clock %s frac@clock %s' % (d['ms_per_step'], d['value']/1e6, r['kernel'], r['achieved'], r['frac'], r['launches'], r['avg_us'], {k: (round(v,3) if isinstance(v,float) else v) for k,v in r.get('all_fp32_gemm_launches',{}).items() if k!='by_variant'}, r.get('sustained_clock_ghz'), r.get('frac_at_sustained_clock')))
print({k: {kk: round(vv,1) for kk,vv in v.items()} for k,v in r['by_variant'].items()})
PY
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --no-clock-probe --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg5: %.4f M env-steps/s  %.2f ms (play %.2f update %.2f)  GEMM %.1f TF/s avg %.1f us' % (d['value']/1e6, d['ms_per_step'], d['play_ms_per_step'], d['update_ms_per_step'], r['achieved'], r['avg_us']))"
