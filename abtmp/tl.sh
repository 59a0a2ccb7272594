#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
for st in 1 0; do
rm -rf $OUT/tl_$st
CAPTRA_L1_STREAM=$st rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$st -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --min-timed-s 0 --min-warmup 2 --no-pose-match --no-cpu-baseline --no-kernel-timing --no-otf --no-b1 --no-legs --mlp-dtype bf16 --lanes 1 > $OUT/tl_$st.log 2>&1
python $ROOT/tools/lane_timeline.py $OUT/tl_$st 1.5 > $OUT/timeline_l1stream$st.txt 2>&1
rm -rf $OUT/tl_$st
done
