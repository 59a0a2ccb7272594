# same-box A/B of the round-4 latency changes (knobs of the library through the environment, captra_amd/_lib.py)
Q="--no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 2 --repeats 5"
OLD="CAPTRA_FPS_DEFER=0 CAPTRA_NN_SPLIT=0 CAPTRA_BQ_CPW=2"
for cfg in "--mlp-dtype bf16" "--mlp-dtype bf16 --batch 16 --lanes 1" "" "--batch 1"; do
  for rep in 1 2; do
    for mode in old new; do
      if [ $mode = old ]; then E="$OLD"; else E=""; fi
      echo -n "[$cfg] $mode: "; env $E python bench.py $Q $cfg 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
    done
  done
done
