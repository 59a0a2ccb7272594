#!/bin/bash
# Same-box A/B of the f32x6 step's late round-6 changes: the dense x6 kernel's store epilogue (CAPTRA_DX_TR=0: straight from the
# accumulators, 1: whole rows through LDS) and split-k of the exact leftovers in the f32x6 mode (CAPTRA_X6_SPLIT_K_POSITIONS).
#   gpurun -- 'bash tools/ab_round6.sh'      -> gpurun_out/ab6/
out=gpurun_out/ab6
mkdir -p $out
for tr in 0 1; do
  for c in 32 16; do CAPTRA_DX_TR=$tr python tools/bench_dense_x6.py --clouds $c >> $out/dense_tr$tr.txt 2>&1; done
done
for cfg in "0 0" "1 0" "0 8192" "1 8192" "1 2048" "1 16384" "0 0" "1 8192"; do
  set -- $cfg
  CAPTRA_DX_TR=$1 CAPTRA_X6_SPLIT_K_POSITIONS=$2 python bench.py --leg --mlp-dtype f32x6 --batch 32 2>>$out/leg.err | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tr=$1 splitk=$2', d['ms_per_step'], d['value'], d.get('pose_match'), d.get('kernel_ms_per_step'))" >> $out/legs.txt
done
cat $out/dense_tr*.txt $out/legs.txt
