#!/bin/bash
# round-6 final bench lines and the full GPU suite (run ON THE GPU BOX): the default command first (cfg2, CPU baseline, clock probe), the same build on the
# 128 x 128 tiling only, then the other configs
mkdir -p gpurun_out/r6
show() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', round(d['ms_per_step'],2), 'ms', round(d['value']), 'env-steps/s', 'play', round(d['play_ms_per_step'],2), 'update', round(d['update_ms_per_step'],2), '| GEMM', round(r['achieved'],1), round(r['frac'],3), 'avg_us', round(r['avg_us'],1), 'clock', r.get('sustained_clock_ghz'), '| cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))"; }
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06_bench_cfg2.json 2> gpurun_out/r6/bench_cfg2_final.err; show gpurun_out/r06_bench_cfg2.json cfg2
PULSE_X3_WIDE=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_cfg2_narrow_tile_only.json 2> gpurun_out/r6/bench_cfg2_narrow.err; show gpurun_out/r06_bench_cfg2_narrow_tile_only.json cfg2_narrow_only
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_cfg2_second.json 2> gpurun_out/r6/bench_cfg2_b.err; show gpurun_out/r06_bench_cfg2_second.json cfg2_again
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r06_bench_cfg5.json 2> gpurun_out/r6/bench_cfg5_final.err; show gpurun_out/r06_bench_cfg5.json cfg5
PULSE_DISC_STREAM=0 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r06_bench_cfg5_one_chain.json 2> gpurun_out/r6/bench_cfg5_oc.err; show gpurun_out/r06_bench_cfg5_one_chain.json cfg5_one_chain
timeout 400 python bench.py --config cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_bench_cfg3.json 2> gpurun_out/r6/bench_cfg3.err; show gpurun_out/r06_bench_cfg3.json cfg3
timeout 300 python bench.py --config terrain_z --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r06_bench_terrain_z.json 2> gpurun_out/r6/bench_tz.err; show gpurun_out/r06_bench_terrain_z.json terrain_z
if [ -z "$SKIP_TESTS" ]; then ( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_gpu_tests.txt; cat gpurun_out/r06_gpu_tests.txt; fi
