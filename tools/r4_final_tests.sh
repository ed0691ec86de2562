#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/full_gpu_tests_final.log 2>&1; tail -6 gpurun_out/r4/full_gpu_tests_final.log
