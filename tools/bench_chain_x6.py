"""Time the f32x6 chain kernels (csrc/chain_x6.hip) next to the exact ones on the bench shapes (32 clouds x 4096 points).
    python tools/bench_chain_x6.py [--clouds 32]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import fused  # noqa: E402
from tools.bench_sa_x6 import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, l = a.clouds, 4096
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, c0, dims in (("chain3 131", 131, [(131, 128), (128, 128), (128, 128)]), ("chain3 134", 134, [(134, 128), (128, 128), (128, 128)]),
                           ("coord tail", 134, [(134, 128), (128, 128), (128, 128), (128, 2), (128, 128), (128, 3)])):
        x = torch.randn(B, c0, l, generator=g).to(dev)
        packed = [fused.pack((torch.randn(*d, generator=g) / np.sqrt(d[0])).to(dev), torch.randn(d[1], generator=g).to(dev)) for d in dims]
        res = {}

        def run(mode):
            with fused.use_mlp_dtype(mode):
                res[mode] = fused.mlp_chain3(x, packed) if len(dims) == 3 else fused.coord_tail(x, packed)

        te, tx = timeit(lambda: run("fp32"), a.iters), timeit(lambda: run("f32x6"), a.iters)
        ya, yb = (res["fp32"], res["f32x6"]) if len(dims) == 3 else (res["fp32"][0], res["f32x6"][0])
        err = float((ya - yb).abs().max() / ya.abs().max())
        macs = B * l * sum(d[0] * d[1] for d in dims)
        print(f"{name}  exact {te:7.1f} us ({2 * macs / te / 1e6:6.1f} TF)   f32x6 {tx:7.1f} us ({2 * macs / tx / 1e6:6.1f} TF-equiv, {6 * 2 * macs / tx / 1e6 / 2500:.3f} of bf16 peak)  "
              f"x{te / tx:.2f}  max err / max |y| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
