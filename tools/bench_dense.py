"""The rotation head's dense layers (Conv -> GroupNorm -> ReLU chain, 128 -> 512 -> 512 -> 256) and plain dense layers at the bench
shapes: time per launch by workgroups per CU (captra_pw_set_occupancy) and with / without paired column tiles.
Usage: python tools/bench_dense.py [--clouds 32]"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import _lib, fused  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    return sum(_lib.prof_read(nm)[0] for nm in _lib.prof_names()) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--points", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    B, L = a.clouds, a.points
    for cin, cout in [(128, 512), (512, 512), (512, 256)]:
        x = torch.randn(B, cin, L, generator=g).to(dev)
        lin = fused.pack((torch.randn(cin, cout, generator=g) / cin ** 0.5).to(dev), torch.randn(cout, generator=g).to(dev))
        ab = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g) * 0.1], -1).to(dev).contiguous()
        fl = 2.0 * B * L * cin * cout
        for label, fn in [("plain + relu      ", lambda: fused.pointwise_mlp(x, lin, fused.ACT_RELU)),
                          ("stats out         ", lambda: fused.pointwise_mlp_gn(x, lin, None, fused.ACT_NONE, True)),
                          ("gn in + stats out ", lambda: fused.pointwise_mlp_gn(x, lin, ab, fused.ACT_NONE, True)),
                          ("gn in             ", lambda: fused.pointwise_mlp_gn(x, lin, ab, fused.ACT_NONE, False))]:
            row = []
            for occ in (4, 3, 2):
                lib.captra_pw_set_occupancy(ctypes.c_int(occ))
                us = timed(fn)
                row.append(f"{occ} wg/CU {us:7.1f} us {fl / us / 1e6:6.1f} TF/s")
            lib.captra_pw_set_occupancy(ctypes.c_int(0))
            print(f"B={B} {cin:3d}->{cout:3d} {label}: " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
