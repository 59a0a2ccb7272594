"""Time the f32x6 SA scales (csrc/sa_x6.hip) next to the exact fp32 kernels on the bench shapes, and print the error against them.
    python tools/bench_sa_x6.py [--clouds 32] [--iters 20]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import fused  # noqa: E402

SHAPES = [(0, (32, 32, 64), 4096, 512, 32), (3, (32, 32, 64), 4096, 512, 32), (0, (64, 64, 128), 4096, 512, 64), (3, (64, 64, 128), 4096, 512, 64),
          (0, (64, 96, 128), 4096, 512, 128), (3, (64, 96, 128), 4096, 512, 128), (320, (128, 128, 256), 512, 128, 64),
          (320, (128, 196, 256), 512, 128, 128)]


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
    B = a.clouds
    for cfeat, chans, n, m, k in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(cfeat + sum(chans))
        xyz_cn = (torch.rand(B, 3, n, generator=g) - 0.5).to(dev)
        feat = torch.randn(B, cfeat, n, generator=g).to(dev) if cfeat else None
        new_xyz = (torch.rand(B, m, 3, generator=g) - 0.5).to(dev)
        idx = torch.randint(0, n, (B, m, k), generator=g, dtype=torch.int32).to(dev)
        dims = (cfeat + 3,) + chans
        packed = [fused.pack((torch.randn(dims[i], dims[i + 1], generator=g) / np.sqrt(dims[i])).to(dev), torch.randn(dims[i + 1], generator=g).to(dev))
                  for i in range(3)]
        out_a = torch.zeros(B, chans[2], m, device=dev)
        out_b = torch.zeros(B, chans[2], m, device=dev)

        def exact():
            if cfeat > 3:
                v1 = fused.sa_first_layer_pre_pm(feat, packed[0])
                fused.sa_scale_pre_pm(v1, xyz_cn, new_xyz, idx, packed, out_a, 0, cfeat)
            else:
                fused.sa_scale_fused(feat, xyz_cn, new_xyz, idx, packed, out_a, 0)

        def x6():
            with fused.use_mlp_dtype("f32x6"):
                fused.sa_scale_x6(feat, xyz_cn, new_xyz, idx, packed, out_b, 0)

        te, tx = timeit(exact, a.iters), timeit(x6, a.iters)
        err = float((out_a - out_b).abs().max() / out_a.abs().max())
        macs = B * m * k * (min(cfeat + 3, 6 if cfeat <= 3 else 3) * chans[0] + chans[0] * chans[1] + chans[1] * chans[2])
        print(f"cf={cfeat:3d} {chans} k={k:3d}  exact {te:8.1f} us ({2 * macs / te / 1e6:6.1f} TF)   f32x6 {tx:8.1f} us ({2 * macs / tx / 1e6:6.1f} TF-equiv, "
              f"{6 * 2 * macs / tx / 1e6 / 2500:.3f} of bf16 peak)   x{te / tx:.2f}   max err / max |y| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
