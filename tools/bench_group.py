"""group_points on the SA shapes (algorithmic bytes 4CN + 4MK + 4CMK per cloud).

Usage: python tools/bench_group.py [--clouds 64]
"""
from __future__ import annotations

import argparse
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from captra_amd import _lib  # noqa: E402
from captra_amd.pointnet_lib import pointnet2_utils as pn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.clouds
    g = torch.Generator(device="cpu").manual_seed(0)
    cases = [("SA1 C=3 K=32", 3, 4096, 512, 32), ("SA1 C=3 K=128", 3, 4096, 512, 128), ("SA1 C=6 K=64", 6, 4096, 512, 64),
             ("SA2 C=320 K=64", 320, 512, 128, 64), ("SA2 C=320 K=128", 320, 512, 128, 128)]
    for name, c, n, m, k in cases:
        feat = torch.randn(B, c, n, generator=g).to(dev)
        idx = torch.randint(0, n, (B, m, k), generator=g, dtype=torch.int32).to(dev)
        nb = B * (4 * c * n + 4 * m * k + 4 * c * m * k)
        if True:
            for _ in range(3):
                out = pn.grouping_operation(feat, idx)
            assert torch.equal(out, torch.gather(feat, 2, idx.long().reshape(B, 1, m * k).expand(-1, c, -1)).reshape(B, c, m, k))
            torch.cuda.synchronize()
            _lib.prof_reset()
            _lib.prof_enable(True)
            for _ in range(args.iters):
                pn.grouping_operation(feat, idx)
            torch.cuda.synchronize()
            _lib.prof_enable(False)
            ms, cnt = _lib.prof_read("group_points")
            per = ms / max(cnt, 1)
            print(f"{name:18s} {per * 1e3:8.1f} us  {nb / per / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
