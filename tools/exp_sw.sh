#!/bin/bash
# EXPERIMENT driver (GPU box): rebuild one source with -D<MACRO>=n and time the SA scales.  n != 0 may give wrong results.
# usage: SRC=sa_pipe.hip MACRO=SP_EXP EXPS="0 1 2 3" ARGS="--pipe --phases" SHAPES="sa2s2" bash tools/exp_sw.sh
cd "$(dirname "$0")/.."
SRC=${SRC:-sa_fused.hip}; MACRO=${MACRO:-SW_EXP}
for e in ${EXPS:-0 1 2 3}; do
  touch captra_amd/csrc/$SRC
  CAPTRA_HIPCC_EXTRA="-D$MACRO=$e $EXTRA" python captra_amd/build.py > /tmp/build_$e.log 2>&1 || { tail -5 /tmp/build_$e.log; exit 1; }
  echo "== $MACRO=$e $EXTRA"
  for w in ${SHAPES:-sa2s2 sa2s1}; do python tools/bench_sa_fused.py --which $w ${ARGS:---pre} --clouds 32 --iters 10 2>/dev/null | grep -v "bit-exact"; done
done
touch captra_amd/csrc/$SRC
