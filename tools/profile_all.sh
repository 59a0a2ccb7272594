#!/bin/bash
# Everything `profiles/` carries for one state of the tree, in one gpurun call:  tools/profile_all.sh <tag>
#   kernel trace + overlap trace + three counter passes for configs[1] (fp32), configs[2]'s arithmetic (bf16), configs[3] (drawers);
#   the same for configs[4] (16384-point backbone); the bench lines of the four; the per-stage benches.
set -u
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
tools/profile_round.sh $TAG
tools/profile_round.sh $TAG _bf16 --mlp-dtype bf16
tools/profile_round.sh $TAG _drawers --category drawers
tools/profile_backbone.sh $TAG
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --leg --mlp-dtype bf16 > $OUT/${TAG}_bench_bf16.json 2> $OUT/${TAG}_bench_bf16.err
python bench.py --leg --category drawers > $OUT/${TAG}_bench_drawers.json 2> $OUT/${TAG}_bench_drawers.err
python tools/bench_backbone.py --npoint 2048 512 > $OUT/${TAG}_backbone16k.json 2> $OUT/${TAG}_backbone16k.err
python tools/bench_dense_bf16.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_dense_bf16.txt
python tools/bench_sa_fused.py --bf16 --clouds 32 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_sa_bf16.txt
python tools/bench_fps.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_fps.txt
python tools/fps_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_fps_ab.txt
python tools/step_breakdown.py --mlp-dtype bf16 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_steps_bf16.txt
python tools/step_breakdown.py --batch 1 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_steps_b1.txt
echo all done
