"""Per-launch breakdown of one tracking step (eager launches): every captra_* call with its shape arguments,
HIP-event time and, for the MFMA layers, achieved TFLOP/s.

    python tools/step_breakdown.py [--batch 32] [--category bottle]
"""
import argparse
import sys
from collections import OrderedDict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import bench  # noqa: E402
from captra_amd import _lib, fused  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mlp-dtype", default="fp32", choices=["fp32", "bf16"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fused.set_mlp_dtype(a.mlp_dtype)
    cfg, sd, model, _ = bench.build_workload(a.batch, dev, mlp_dtype=a.mlp_dtype)
    pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
    records = OrderedDict()
    orig_call = _lib.call
    state = {"on": False, "seq": 0}

    def timed_call(name, *args):
        if not state["on"]:
            return orig_call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_call(name, *args)
        e1.record()
        ints = tuple(x for x in args if isinstance(x, int) and abs(x) < (1 << 24))
        key = (state["seq"], name, ints)
        state["seq"] += 1
        records.setdefault(key, []).append((e0, e1))
        return r

    _lib.call = timed_call
    fused.L.call = timed_call
    with torch.no_grad():
        for rep in range(a.reps + 1):
            state["on"] = rep > 0
            state["seq"] = 0
            _, pose = model.track_step(model.feed_dict[1], model.npcs_feed_dict[1], pose)
    torch.cuda.synchronize()
    tot = 0.0
    print(f"{'#':>3s} {'entry point':28s} {'int args':44s} {'us':>9s} {'TFLOP/s':>8s}")
    for (seq, name, ints), evs in records.items():
        us = 1e3 * sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
        tot += us
        tf = ""
        if name in ("captra_pointwise_mlp", "captra_pointwise_mlp_bf16", "captra_pointwise_mlp_bf16_pm", "captra_pointwise_mlp_pm", "captra_pointwise_mlp_gn"):
            b, cin, cout, l = ints[:4]
            tf = f"{2.0 * b * cin * cout * l / us / 1e6:8.1f}"
        elif name in ("captra_sa_scale_fused", "captra_sa_scale_bf16", "captra_sa_scale_pre", "captra_sa_scale_pre_pm"):
            b, n, m, k, cf, c1, c2, c3 = ints[:8]
            cin1 = 3 if (name != "captra_sa_scale_fused" and cf > 3) else cf + 3      # pre-transformed first layer: xyz rows only
            tf = f"{2.0 * b * m * k * (cin1 * c1 + c1 * c2 + c2 * c3) / us / 1e6:8.1f}"
        elif name == "captra_mlp_chain3":
            b, c0, c1, c2, c3, l = ints[:6]
            tf = f"{2.0 * b * l * (c0 * c1 + c1 * c2 + c2 * c3) / us / 1e6:8.1f}"
        elif name == "captra_mlp_max":
            b, cin, cout, m, k = ints[:5]
            tf = f"{2.0 * b * cin * cout * m * k / us / 1e6:8.1f}"
        print(f"{seq:3d} {name:28s} {str(ints):44s} {us:9.1f} {tf:>8s}")
    print(f"sum of bracketed launches: {tot / 1e3:.3f} ms (event brackets include launch gaps)")


if __name__ == "__main__":
    main()
