#!/bin/bash
# rocprofv3 kernel trace of the drop-in group_points / ball_query ops on the SA shapes (32 clouds) beside the tool's own event timing
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bench_group.py --clouds 32 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_group.txt
python tools/bench_ball_query.py 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_group.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_group_trace -o g -- python $ROOT/tools/bench_group.py --clouds 32 > $OUT/${TAG}_group_trace.log 2>&1
DB=$(ls $OUT/${TAG}_group_trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_group_kernel_trace_stats.txt 2>&1
rm -rf $OUT/${TAG}_group_trace
cd $ROOT
