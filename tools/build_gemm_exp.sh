#!/bin/bash
# timing-experiment builds of the GEMM (PULSE_GEMM_EXP bit flags, wrong results by design): tools/gemm_bench_exp<N>
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipulse_amd/csrc -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DPULSE_GEMM_EXP=$n $XFLAGS tools/gemm_bench.cpp \
      -x hip pulse_amd/csrc/gemm_f32.hip pulse_amd/csrc/capi.cpp -o tools/gemm_bench_exp$n &
done
wait
