Q="--mlp-dtype bf16 --no-cpu-baseline --no-otf --no-b1 --no-legs --no-pose-match --no-kernel-timing --min-timed-s 1 --repeats 3"
for v in "" "--lanes 1" "--lanes 1 --no-overlap" "--batch 16 --lanes 1" "--lanes 3" "--lanes 4" "--batch 64" "--batch 64 --lanes 4" "--batch 48 --lanes 3"; do
  echo "== $v"; python bench.py $Q $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
done
