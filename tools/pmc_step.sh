#!/bin/bash
# SQ counters of one eager tracking step (tools/step_breakdown.py) per kernel: tools/pmc_step.sh <tag> [step_breakdown args] [-- kernel-name filter]
set -u
TAG=${1:-pmc}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA \
    --kernel-trace --output-format csv -d $OUT/${TAG}_sq -o p -- python $ROOT/tools/step_breakdown.py --reps 2 "$@" > $OUT/${TAG}_sq.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_sq --json $OUT/${TAG}_step_pmc.json > /dev/null 2>&1
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_step_pmc.json"))["kernels"]
print(f"{'kernel':72s} {'us':>7s} {'n':>3s} {'mfma%':>6s} {'wait%':>6s} {'istall%':>7s} {'act%':>5s} {'waves/simd':>10s}")
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("us_in_sq_pass", {}).get("mean", 0) * kv[1].get("us_in_sq_pass", {}).get("dispatches", 0)):
    if "SQ_WAVE_CYCLES" not in c or c["us_in_sq_pass"]["mean"] < 8:
        continue
    us = c["us_in_sq_pass"]["mean"]; cyc = us * 1e-6 * 2.1e9 * 1024; wc = c["SQ_WAVE_CYCLES"]["mean"] * 4
    print(f"{k[:72]:72s} {us:7.1f} {c['us_in_sq_pass']['dispatches']:3d} {c['SQ_VALU_MFMA_BUSY_CYCLES']['mean'] / cyc * 100:6.1f} "
          f"{c['SQ_WAIT_ANY']['mean'] * 4 / wc * 100:6.1f} {c['SQ_WAIT_INST_ANY']['mean'] * 4 / wc * 100:7.1f} {c['SQ_ACTIVE_INST_ANY']['mean'] * 4 / wc * 100:5.1f} {wc / cyc:10.2f}")
PY
