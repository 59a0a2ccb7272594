#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the 16384-point backbone workload
# (BASELINE.json configs[4]).   tools/profile_backbone.sh <tag>
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_backbone.py --npoint 2048 512 --steps 5 --warmup 2 --no-kernel-timing --no-pipe"
CMD_EAGER="python $ROOT/tools/bench_backbone.py --npoint 2048 512 --steps 2 --warmup 1 --no-kernel-timing --no-graph"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_bb_trace -o bench -- $CMD > $OUT/${TAG}_bb_trace.log 2>&1
DB=$(ls $OUT/${TAG}_bb_trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_backbone16k_kernel_trace_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_bb_fetch -o p -- $CMD_EAGER > $OUT/${TAG}_bb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_bb_write -o p -- $CMD_EAGER > $OUT/${TAG}_bb_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA \
    --kernel-trace --output-format csv -d $OUT/${TAG}_bb_sq -o p -- $CMD_EAGER > $OUT/${TAG}_bb_sq.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_bb_fetch $OUT/${TAG}_bb_write $OUT/${TAG}_bb_sq --json $OUT/${TAG}_backbone16k_pmc.json > $OUT/${TAG}_backbone16k_pmc_summary.txt 2>&1
# the raw rocprofv3 output (rocpd databases, counter CSVs) is tens of MiB per pass: dropped unless KEEP_RAW=1 (gpurun merges <= 64 MiB back)
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf $OUT/${TAG}_bb_trace $OUT/${TAG}_bb_trace_overlap $OUT/${TAG}_bb_fetch $OUT/${TAG}_bb_write $OUT/${TAG}_bb_sq
echo done
