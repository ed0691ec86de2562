#!/bin/bash
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_b16_ops_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -5 gpurun_out/r4/t_b16_ops.log
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/prof5 -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe > $ROOT/gpurun_out/r4/bench_cfg5_b16_profiled.json 2> $ROOT/gpurun_out/r4/prof5.err
DB=$(find $ROOT/gpurun_out/r4/prof5 -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4/kstats_cfg5_b16.md | head -40
rm -rf $ROOT/gpurun_out/r4/prof5
