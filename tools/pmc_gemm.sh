#!/bin/bash
# SQ / GRBM counters for the GEMM kernels on the cfg2 layer-1 forward shape and 4096^3 (run ON THE GPU BOX):
#   tools/pmc_gemm.sh <out-file>
# One --pmc pass per counter group (SQ: 8 slots, GRBM: 2), kernel trace only (gpurun refuses --pmc mixed with other trace domains).
set -u
OUTF=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_gemm
rm -rf $D; mkdir -p $D
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
G2="GRBM_GUI_ACTIVE GRBM_COUNT"
G3="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -d $D/p$i --output-format csv -- python $ROOT/tools/pmc_gemm_driver.py > /dev/null 2> $D/p$i.err
done
python $ROOT/tools/pmc_gemm_report.py $D > $OUTF
tail -3 $D/p1.err >> $OUTF
