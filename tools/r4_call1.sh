#!/bin/bash
# round 4, first GPU call: feed probe, self-launch bench test, baselines of cfg2 / cfg5 on this box
mkdir -p gpurun_out/r4
tools/mfma_feed_probe > gpurun_out/r4/mfma_feed_probe.txt 2>&1
timeout 900 python -m pytest tests/test_bench_launch_gpu.py -x -q > gpurun_out/r4/test_bench_launch.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r4/bench_cfg2_base.json 2> gpurun_out/r4/bench_cfg2_base.err
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/r4/bench_cfg5_base.json 2> gpurun_out/r4/bench_cfg5_base.err
tail -3 gpurun_out/r4/test_bench_launch.log; cat gpurun_out/r4/mfma_feed_probe.txt
