#!/bin/bash
python -m pytest tests/test_model_gpu.py -q -x -k "bf16_sa_scale" 2>&1 | tail -3
for v in 8 0; do for c in 32 16; do echo "== variant $v clouds $c"; CAPTRA_SA_BF16_VARIANT=$v python tools/bench_sa_fused.py --bf16 --clouds $c 2>&1 | grep -v amdgpu.ids | grep sa2; done; done
Q="--mlp-dtype bf16 --no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 2 --repeats 5"
for rep in 1 2; do for v in 8 0; do echo -n "bf16 step variant $v: "; CAPTRA_SA_BF16_VARIANT=$v python bench.py $Q 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"; done; done
