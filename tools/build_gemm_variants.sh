#!/bin/bash
# Macro variants of gemm_f32.hip linked into separate copies of the library (build/variants/libpulse_gemm_<name>.so) for A/B timing in ONE
# gpurun call (boxes differ by several %):   tools/build_gemm_variants.sh name1 "-DX3_CSTRIDE=136" name2 "-DX3_BARRIER_GAP=9" ...
# then on the box:   PULSE_HIP_LIB=build/variants/libpulse_gemm_<name>.so python bench.py --no-cpu-baseline
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pulse_amd/csrc
OUT=$ROOT/build/variants
mkdir -p "$OUT"
OBJS=$(ls $SRC/*.o | grep -v gemm_f32.o | tr '\n' ' ')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -fno-slp-vectorize -fno-vectorize -mllvm -amdgpu-mfma-vgpr-form $flags -c $SRC/gemm_f32.hip -o $OUT/gemm_f32_$name.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libpulse_gemm_$name.so $OBJS $OUT/gemm_f32_$name.o
    echo "built $name ($flags)" ) &
done
wait
