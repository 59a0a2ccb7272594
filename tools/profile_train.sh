#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/r01i_train_trace -o bench -- python $ROOT/tools/bench_train.py --steps 5 --warmup 2 > $OUT/r01i_train_trace.log 2>&1
DB=$(ls $OUT/r01i_train_trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_summary.py $DB --top 45 > $OUT/r01i_train_kernel_trace_stats.txt 2>&1
echo done
