#!/bin/bash
mkdir -p gpurun_out/r4
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 400 python bench.py --config cfg3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r4/r04_bench_cfg3.json 2> gpurun_out/r4/bench_cfg3.err; python -c "
import json; d=json.loads(open('gpurun_out/r4/r04_bench_cfg3.json').read().strip().splitlines()[-1]); print('cfg3', round(d['ms_per_step'],1), round(d['value']), round(d['roofline']['achieved'],1), d['roofline']['frac'])"
timeout 300 python bench.py --config terrain_z --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/r04_bench_terrain_z.json 2> gpurun_out/r4/bench_tz.err; python -c "
import json; d=json.loads(open('gpurun_out/r4/r04_bench_terrain_z.json').read().strip().splitlines()[-1]); print('terrain_z', round(d['ms_per_step'],1), round(d['value']), round(d['roofline']['achieved'],1))"
timeout 300 python bench.py --config cfg5_f32 --no-cpu-baseline --steps 3 --warmup 1 --no-clock-probe > gpurun_out/r4/r04_bench_cfg5_f32.json 2> gpurun_out/r4/bench_c5f.err; python -c "
import json; d=json.loads(open('gpurun_out/r4/r04_bench_cfg5_f32.json').read().strip().splitlines()[-1]); print('cfg5_f32', round(d['ms_per_step'],1), round(d['value']), round(d['roofline']['achieved'],1))"
