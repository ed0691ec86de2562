#!/bin/bash
# builds tools/gemm_bench against the in-tree libpulse_hip.so (run pulse_amd/csrc/build.py first)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lpulse_amd/csrc -lpulse_hip \
    -Wl,-rpath,'$ORIGIN/../pulse_amd/csrc' -o tools/gemm_bench
