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
    ap.add_argument("--sweep", action="store_true", help="staging budget / channel chunk / positions per workgroup of the kernel (captra_group_set_shape)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.clouds
    g = torch.Generator(device="cpu").manual_seed(0)
    cases = [("SA1 C=3 K=32", 3, 4096, 512, 32), ("SA1 C=3 K=128", 3, 4096, 512, 128), ("SA1 C=6 K=64", 6, 4096, 512, 64),
             ("SA2 C=320 K=64", 320, 512, 128, 64), ("SA2 C=320 K=128", 320, 512, 128, 128)]
    shapes = [(64, 32, 0)]
    if args.sweep:
        shapes += [(64, 32, 8192), (64, 16, 0), (32, 16, 0), (32, 16, 8192), (48, 24, 0), (16, 8, 0), (16, 8, 8192), (32, 8, 0), (64, 32, 16384)]
        probe = torch.empty(128 << 20, dtype=torch.float32, device=dev)
        probe.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            probe.fill_(2.0)
        e1.record()
        torch.cuda.synchronize()
        print(f"fill probe {5 * probe.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9:.0f} GB/s")
        del probe
    for shp in shapes:
      _lib.lib().captra_group_set_shape(*[ctypes.c_int(v) for v in shp])
      if args.sweep:
          print(f"-- staging {shp[0]} KiB, <= {shp[1]} channels per workgroup, positions per workgroup {shp[2] or 'default'}")
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
