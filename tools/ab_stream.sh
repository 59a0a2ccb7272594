#!/bin/bash
# streamed level-1 sampling (CAPTRA_SAMPLER_CHUNKS) against the plain step, same box: fp32 / bf16, one lane / two lanes, batch 1
Q="--no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 2 --repeats 5"
run() { echo -n "$1 chunks=$2 [$3]: "; CAPTRA_SAMPLER_CHUNKS=$2 python bench.py $Q $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"; }
for ch in 0 2 4 8; do run bf16 $ch "--mlp-dtype bf16 --lanes 1"; done
for ch in 0 2 4; do run bf16 $ch "--mlp-dtype bf16 --lanes 2"; done
for ch in 0 2 4 8; do run fp32 $ch "--lanes 1"; done
for ch in 0 4; do run fp32 $ch "--lanes 2"; done
for ch in 0 2 4 8; do run b1 $ch "--batch 1"; done
