#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_rccl_world1_gpu.py tests/test_bf16_gpu.py "tests/test_full_size_parity_gpu.py::test_cfg5_full_size_first_minibatch_bf16" tests/test_agent_parity2_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r5c5_tests.txt
cat gpurun_out/r5c5_tests.txt
