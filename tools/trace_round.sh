#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/t1_trace -o bench -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/t1_trace.log 2>&1
cd $ROOT
python tools/rocpd_summary.py $(ls $OUT/t1_trace/*.db | head -1) --step-trace 1 > $OUT/t1_stats.txt 2>&1
