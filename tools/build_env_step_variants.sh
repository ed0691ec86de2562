#!/bin/bash
# Compile-flag variants of env_step.hip linked into separate copies of the library (build/variants/libpulse_<name>.so), for
# tools/im_step_repro.py --lib (bisecting the contention-dependent output of the fused env step, round 3).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pulse_amd/csrc
OUT=$ROOT/build/variants
mkdir -p "$OUT"
OBJS=$(ls $SRC/*.o | grep -v env_step.o | tr '\n' ' ')
one() {
  name=$1; shift
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -ffp-contract=off "$@" -c $SRC/env_step.hip -o $OUT/env_step_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libpulse_$name.so $OBJS $OUT/env_step_$name.o
  echo "built $name"
}
one noslp -fno-slp-vectorize &
one o1 -O1 &
one nospillv -mllvm -amdgpu-spill-sgpr-to-vgpr=0 &
one wait0 -mllvm -amdgpu-waitcnt-forcezero &
wait
