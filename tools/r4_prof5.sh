#!/bin/bash
# kernel table + one-minibatch / one-step windows of cfg5 (run ON THE GPU BOX): tools/r4_prof5.sh <tag>
TAG=$1
mkdir -p gpurun_out/r4
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/prof5 -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe > $ROOT/gpurun_out/r4/bench_cfg5_${TAG}_profiled.json 2> $ROOT/gpurun_out/r4/prof5.err
DB=$(find $ROOT/gpurun_out/r4/prof5 -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe   (cfg5, MI355X, r04 $TAG)" > $ROOT/gpurun_out/r4/kstats_cfg5_${TAG}.md
echo >> $ROOT/gpurun_out/r4/kstats_cfg5_${TAG}.md
python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4/kstats_cfg5_${TAG}.md > /dev/null
python $ROOT/tools/rocprof_window.py "$DB" kinematic_sim_kernel 50 > $ROOT/gpurun_out/r4/window_step_cfg5_${TAG}.txt
python $ROOT/tools/rocprof_window.py "$DB" ppo_loss_kernel 120 > $ROOT/gpurun_out/r4/window_mb_cfg5_${TAG}.txt
rm -rf $ROOT/gpurun_out/r4/prof5
