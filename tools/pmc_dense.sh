#!/bin/bash
# PMC passes over tools/bench_dense_bf16.py: tools/pmc_dense.sh <tag>   -> gpurun_out/<tag>_dense_pmc.txt
set -u
TAG=${1:-pmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_dense_bf16.py --iters 3 ${2:-}"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
   --kernel-trace --output-format csv -d $OUT/${TAG}_dense_a -o p -- $CMD > $OUT/${TAG}_dense_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM \
   --kernel-trace --output-format csv -d $OUT/${TAG}_dense_b -o p -- $CMD > $OUT/${TAG}_dense_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $OUT/${TAG}_dense_c -o p -- $CMD > $OUT/${TAG}_dense_c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_dense_d -o p -- $CMD > $OUT/${TAG}_dense_d.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_dense_e -o p -- $CMD > $OUT/${TAG}_dense_e.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_dense_a $OUT/${TAG}_dense_b $OUT/${TAG}_dense_c $OUT/${TAG}_dense_d $OUT/${TAG}_dense_e | grep -E "^kernel|tb_|pw_bf16pm" > $OUT/${TAG}_dense_pmc.txt
rm -rf $OUT/${TAG}_dense_[a-e]
