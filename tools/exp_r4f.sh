#!/bin/bash
Q="--no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 3 --repeats 5"
for rep in 1 2; do for v in "" 2; do echo -n "fp32 step CAPTRA_SA_SPLIT=$v: "; if [ -z "$v" ]; then python bench.py $Q 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"; else CAPTRA_SA_SPLIT=$v python bench.py $Q 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"; fi; done; done
echo "lanes 1:"; python bench.py $Q --lanes 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
for v in "" 2; do echo "sa_fused bench split=$v"; CAPTRA_SA_SPLIT=$v python tools/bench_sa_fused.py --clouds 16 --pipe 2>&1 | grep -v amdgpu.ids; done
