#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
python -m pytest tests/test_model_gpu.py -q -x -k "bf16_sa_scale" 2>&1 | tail -3
for v in 0 16 32 48 64 112; do for c in 32 16; do echo "== ablation $((v/16)) clouds $c"; CAPTRA_SA_BF16_VARIANT=$((v+1)) python tools/bench_sa_fused.py --bf16 --clouds $c --which sa2s2 2>&1 | grep -v amdgpu.ids;  CAPTRA_SA_BF16_VARIANT=$((v+1)) python tools/bench_sa_fused.py --bf16 --clouds $c --which sa1s3x 2>&1 | grep -v amdgpu.ids; done; done
python bench.py --no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --min-timed-s 2 --repeats 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('hbm_ops'),indent=1))"
