#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the bench workload.
#   tools/profile_round.sh <tag> [<suffix> [extra bench.py args …]]
#       -> gpurun_out/<tag><suffix>_{trace,fetch,write,sq}/…  + text summaries gpurun_out/<tag>_bench<suffix>_*
#   e.g. tools/profile_round.sh r03a _bf16 --mlp-dtype bf16      tools/profile_round.sh r03a _drawers --category drawers
set -u
TAG=${1:-r01}
SUF=${2:-}
shift; shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# --no-overlap: the two networks one after the other, so that a dispatch's duration is the kernel's own (bench.py's roofline
# figures are measured the same way); the default bench line runs them side by side on two streams
QUICK="--repeats 1 --min-timed-s 0 --min-warmup 2 --no-pose-match --no-cpu-baseline --no-kernel-timing --no-otf --no-b1 $EXTRA"
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 $QUICK --no-overlap --lanes 1"
BENCH_EAGER="python $ROOT/bench.py --steps 2 --warmup 1 $QUICK --no-graph --no-overlap"
BENCH_OVERLAP="python $ROOT/bench.py --steps 5 --warmup 2 $QUICK"
cd /tmp
# 1. kernel trace + stats of the bench command (hipGraph replay path)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}${SUF}_trace -o bench -- $BENCH > $OUT/${TAG}${SUF}_trace.log 2>&1
DB=$(ls $OUT/${TAG}${SUF}_trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_summary.py $DB --step-trace 1 > $OUT/${TAG}_bench${SUF}_kernel_trace_stats.txt 2>&1
# 1b. the default command (networks side by side): kernel trace only
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}${SUF}_trace_overlap -o bench -- $BENCH_OVERLAP > $OUT/${TAG}${SUF}_trace_overlap.log 2>&1
DB2=$(ls $OUT/${TAG}${SUF}_trace_overlap/*.db 2>/dev/null | head -1)
[ -n "$DB2" ] && python $ROOT/tools/rocpd_summary.py $DB2 > $OUT/${TAG}_bench${SUF}_overlap_kernel_trace_stats.txt 2>&1
# 2. PMC passes (separate runs, counters only; eager launches so every dispatch is a kernel node the tool sees)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}${SUF}_fetch -o p -- $BENCH_EAGER > $OUT/${TAG}${SUF}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}${SUF}_write -o p -- $BENCH_EAGER > $OUT/${TAG}${SUF}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA \
    --kernel-trace --output-format csv -d $OUT/${TAG}${SUF}_sq -o p -- $BENCH_EAGER > $OUT/${TAG}${SUF}_sq.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}${SUF}_fetch $OUT/${TAG}${SUF}_write $OUT/${TAG}${SUF}_sq --json $OUT/${TAG}_bench${SUF}_pmc.json > $OUT/${TAG}_bench${SUF}_pmc_summary.txt 2>&1
# the raw rocprofv3 output (rocpd databases, counter CSVs) is tens of MiB per pass: dropped unless KEEP_RAW=1 (gpurun merges <= 64 MiB back)
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf $OUT/${TAG}${SUF}_trace $OUT/${TAG}${SUF}_trace_overlap $OUT/${TAG}${SUF}_fetch $OUT/${TAG}${SUF}_write $OUT/${TAG}${SUF}_sq
echo done
