#!/bin/bash
# PMC passes over tools/bench_sa_fused.py for one SA shape: tools/pmc_sa.sh <shape> <tag>
set -u
SHAPE=${1:-sa1s3}; TAG=${2:-pmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_sa_fused.py --which $SHAPE --clouds 32 --iters 3"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM \
   --kernel-trace --output-format csv -d $OUT/${TAG}_${SHAPE}_a -o p -- $CMD > $OUT/${TAG}_${SHAPE}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA \
   --kernel-trace --output-format csv -d $OUT/${TAG}_${SHAPE}_b -o p -- $CMD > $OUT/${TAG}_${SHAPE}_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $OUT/${TAG}_${SHAPE}_c -o p -- $CMD > $OUT/${TAG}_${SHAPE}_c.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_${SHAPE}_a $OUT/${TAG}_${SHAPE}_b $OUT/${TAG}_${SHAPE}_c | grep -E "^kernel|sa_wave|sa_fused" > $OUT/${TAG}_${SHAPE}.txt
