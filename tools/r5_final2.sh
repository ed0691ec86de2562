#!/bin/bash
# round-5 closing run (ON THE GPU BOX): the default bench command on the final build (new roofline layout), then the full GPU suite
mkdir -p gpurun_out/r5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05_bench_cfg2.json 2> gpurun_out/r5/bench_cfg2_final2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_cfg2.json')); r=d['roofline']; a=r.get('all_fp32_gemm_launches', {})
print('cfg2 %.2f ms %.4f M env-steps/s play %.2f update %.2f | %s %.1f TF/s frac %.3f avg %.1f us x %d | all x3 %.1f (%.3f) | clock %.2f frac@clock %.3f | traffic %.1f MB (%s) | cpu %.0f on %s threads -> %.0fx' % (
    d['ms_per_step'], d['value']/1e6, d['play_ms_per_step'], d['update_ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_us'], r['launches'], a.get('achieved', 0), a.get('frac', 0),
    r.get('sustained_clock_ghz') or 0, r.get('frac_at_sustained_clock') or 0, (r.get('traffic') or 0)/1e6, r.get('traffic_source'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d.get('speedup_vs_cpu', 0)))
PY
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r05_gpu_tests.txt; cat gpurun_out/r05_gpu_tests.txt
