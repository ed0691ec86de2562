#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_b16_ops_gpu.py tests/test_learner_kernels_gpu.py tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py tests/test_disc_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -4 gpurun_out/r4/t_b16_ops.log
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/prof5 -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe > $ROOT/gpurun_out/r4/bench_cfg5_b16_profiled.json 2> $ROOT/gpurun_out/r4/prof5.err
DB=$(find $ROOT/gpurun_out/r4/prof5 -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4/kstats_cfg5_b16_v6.md > /dev/null
python $ROOT/tools/rocprof_window.py "$DB" kinematic_sim_kernel 50 > $ROOT/gpurun_out/r4/window_step_cfg5.txt
python $ROOT/tools/rocprof_window.py "$DB" ppo_loss_kernel 120 > $ROOT/gpurun_out/r4/window_mb_cfg5.txt
rm -rf $ROOT/gpurun_out/r4/prof5
cd $ROOT
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_b16_v6.json 2> gpurun_out/r4/bench_cfg5_b16_v6.err
