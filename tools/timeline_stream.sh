#!/bin/bash
# kernel timeline of ONE lane's step with the streamed sampler (one graph, three branches): does the sampler overlap the windows?
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
for ch in 0 2; do
rm -rf $OUT/tl_$ch
CAPTRA_SAMPLER_CHUNKS=$ch rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$ch -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --min-timed-s 0 --min-warmup 2 --no-pose-match --no-cpu-baseline --no-kernel-timing --no-otf --no-b1 --no-legs --mlp-dtype bf16 --lanes 1 > $OUT/tl_$ch.log 2>&1
python $ROOT/tools/lane_timeline.py $OUT/tl_$ch 1.6 > $OUT/timeline_chunks$ch.txt 2>&1
rm -rf $OUT/tl_$ch
done
