run() { CAPTRA_DEFER_FP1=0 timeout 300 python bench.py --leg --mlp-dtype bf16 --batch 32 "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d.get('l1_stream'))
"; }
for rep in 1 2; do
for g in 128 192 256 320 384; do echo "both lanes2 grid$g"; CAPTRA_L1_GRID=$g CAPTRA_L1_NETS=both run --lanes 2; done
echo "off lanes2"; CAPTRA_L1_STREAM=0 run --lanes 2
done
echo B64; for g in 256 384 512; do echo "both lanes2 grid$g"; CAPTRA_L1_GRID=$g CAPTRA_L1_NETS=both run --lanes 2 --batch 64; done
echo "off lanes2"; CAPTRA_L1_STREAM=0 run --lanes 2 --batch 64
