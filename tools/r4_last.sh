#!/bin/bash
# last checks of round 4 (run ON THE GPU BOX): the tests around the last change, a cfg5 bench line, the steady-state dispatch count
mkdir -p gpurun_out/r4 gpurun_out/r4p gpurun_out/prof_cfg5
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_b16_ops_gpu.py tests/test_disc_gpu.py tests/test_amp_agent_gpu.py tests/test_bf16_gpu.py tests/test_agent_parity2_gpu.py -q > gpurun_out/r4/t_last.log 2>&1; tail -3 gpurun_out/r4/t_last.log
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r04_bench_cfg5.json 2> gpurun_out/r4/bench_cfg5_final.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_cfg5.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5', round(d['ms_per_step'],2), round(d['value']), d['per_step_play_update_ms'], round(r['achieved'],1))"
for K in 2 5; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_cfg5/trace$K -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps $K --warmup 1 --no-clock-probe > /dev/null 2> $ROOT/gpurun_out/r4p/trace$K.err
  DB=$(find $ROOT/gpurun_out/prof_cfg5/trace$K -name "*.db" | head -1); rm -f $ROOT/gpurun_out/r4p/kstats_cfg5_$K.md; python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4p/kstats_cfg5_$K.md > /dev/null; rm -rf $ROOT/gpurun_out/prof_cfg5/trace$K )
done
python - <<'PY'
import re
def disp(p):
    return int(re.search(r"(\d+) dispatches", open(p).read()).group(1))
a, b = disp("gpurun_out/r4p/kstats_cfg5_2.md"), disp("gpurun_out/r4p/kstats_cfg5_5.md")
open("gpurun_out/r04_dispatches_cfg5.txt", "w").write(
    f"# cfg5 dispatches (rocprofv3 --kernel-trace): {a} over warm-up + 2 epochs (includes the one-time AMP demo-buffer fill: 391 fetches x ~17 launches),\n"
    f"# {b} over warm-up + 5 epochs -> steady state {(b - a) / 3:.0f} dispatches per epoch\n")
print(open("gpurun_out/r04_dispatches_cfg5.txt").read())
PY
