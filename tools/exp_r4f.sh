#!/bin/bash
Q="--no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 3 --repeats 5"
for rep in 1 2; do for v in 0 1; do echo -n "fp32 step CAPTRA_SA_PRIO=$v: "; CAPTRA_SA_PRIO=$v python bench.py $Q 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"; done; done
for v in 0 1; do echo "pipe prio $v"; CAPTRA_SA_PRIO=$v python tools/exp_pipe_stages.py 2>&1 | grep -v amdgpu.ids | tail -3; done
