"""Time the f32x6 dense layers of the rotation heads (csrc/dense_x6.hip) next to the exact fp32 kernels (captra_pointwise_mlp_gn).
    python tools/bench_dense_x6.py [--clouds 32] [--iters 20]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import fused  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L = a.clouds, 4096
    for cin, cout, gn_in in ((128, 512, False), (512, 512, True), (512, 256, True)):
        g = torch.Generator(device="cpu").manual_seed(cin + cout)
        x = torch.randn(B, cin, L, generator=g).to(dev)
        lin = fused.pack((torch.randn(cin, cout, generator=g) / np.sqrt(cin)).to(dev), torch.randn(cout, generator=g).to(dev))
        ab = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g) * 0.3], dim=-1).to(dev) if gn_in else None
        res = {}
        for mode in ("fp32", "f32x6"):
            def run():
                with fused.use_mlp_dtype(mode):
                    return fused.pointwise_mlp_gn(x, lin, ab, fused.ACT_NONE, want_stats=True)
            res[mode] = (timeit(run, a.iters), run()[0])
        (te, ye), (tx, yx) = res["fp32"], res["f32x6"]
        fl = 2.0 * B * L * cin * cout
        print(f"{cin:4d} -> {cout:4d} (GroupNorm on load: {gn_in})  exact {te:7.1f} us ({fl / te / 1e6:6.1f} TF)   f32x6 {tx:7.1f} us ({fl / tx / 1e6:6.1f} TF-equiv, "
              f"{6 * fl / tx / 1e6 / 2500:.3f} of bf16 peak)  x{te / tx:.2f}  max err / max |y| = {float((ye - yx).abs().max() / ye.abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
