#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_b16_ops_gpu.py -x -q > gpurun_out/r4/t_b16_ops.log 2>&1; tail -5 gpurun_out/r4/t_b16_ops.log
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_disc_gpu.py tests/test_amp_agent_gpu.py -q -x > gpurun_out/r4/t_bf16.log 2>&1; tail -5 gpurun_out/r4/t_bf16.log
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/prof5 -- python $ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 --no-clock-probe > $ROOT/gpurun_out/r4/bench_cfg5_b16_profiled.json 2> $ROOT/gpurun_out/r4/prof5.err
DB=$(find $ROOT/gpurun_out/r4/prof5 -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$DB" $ROOT/gpurun_out/r4/kstats_cfg5_b16_v3.md > /dev/null
rm -rf $ROOT/gpurun_out/r4/prof5
cd $ROOT
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_b16_v3.json 2> gpurun_out/r4/bench_cfg5_b16_v3.err
PMC_DRIVER=b16 bash tools/pmc_gemm.sh $ROOT/gpurun_out/r4/pmc_b16.txt
