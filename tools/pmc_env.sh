#!/bin/bash
# SQ / GRBM / TA counters for the fused env step and the terrain step (run ON THE GPU BOX):   tools/pmc_env.sh <out-file>
set -u
OUTF=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_env
rm -rf $D; mkdir -p $D
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
G2="GRBM_GUI_ACTIVE GRBM_COUNT"
G3="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -d $D/p$i --output-format csv -- python $ROOT/tools/pmc_env_driver.py > /dev/null 2> $D/p$i.err
done
PMC_KERNEL_FILTER="im_step traj_step" python $ROOT/tools/pmc_gemm_report.py $D > $OUTF
# (round 6: the rocprofv3 log tail is no longer appended to the counter file)
