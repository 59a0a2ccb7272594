"""Per-kernel timings of the geometry operators at the bench workload (B clouds of N points).

Usage: python tools/microbench.py [--clouds 64] [--iters 20]
Prints one line per kernel: average device time from the library's own HIP-event profiler, the
algorithmic bytes of SURVEY.md §8(d) and the implied GB/s.
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
from captra_amd import synthetic as clouds  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fps-waves", type=int, nargs="*", default=[0, 1, 2, 4, 8, 16])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, N = args.clouds, 4096
    base = np.stack([clouds.s_nocs(i)[0] for i in range(8)])
    xyz = torch.from_numpy(np.tile(base, (B // 8 + 1, 1, 1))[:B]).to(dev).contiguous()

    def timed(name, fn, iters=args.iters, nbytes=None, extra=""):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        ms, n = _lib.prof_read(name)
        per = ms / max(n, 1)
        bw = f"{nbytes / per / 1e6:9.1f} GB/s" if nbytes else ""
        print(f"{name:18s} {extra:28s} {per * 1e3:10.1f} us/launch  ({n} launches) {bw}", flush=True)
        return per

    # ---- FPS
    for w in args.fps_waves:
        _lib.lib().captra_fps_set_waves(ctypes.c_int(w))
        timed("fps", lambda: pn.furthest_point_sample(xyz, 512), extra=f"N=4096 M=512 waves={w}")
    _lib.lib().captra_fps_set_waves(ctypes.c_int(0))
    idx1 = pn.furthest_point_sample(xyz, 512)
    xyz1 = torch.gather(xyz, 1, idx1.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    for w in [0, 1, 2, 4, 8]:
        _lib.lib().captra_fps_set_waves(ctypes.c_int(w))
        timed("fps", lambda: pn.furthest_point_sample(xyz1, 128), extra=f"N=512 M=128 waves={w}")
    _lib.lib().captra_fps_set_waves(ctypes.c_int(0))
    idx2 = pn.furthest_point_sample(xyz1, 128)
    xyz2 = torch.gather(xyz1, 1, idx2.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()

    # ---- ball query (algorithmic bytes 12N + 12M + 4MK per cloud)
    for (r, k) in [(0.05, 32), (0.1, 64), (0.2, 128)]:
        nb = B * (12 * N + 12 * 512 + 4 * 512 * k)
        timed("ball_query", lambda: pn.ball_query(r, k, xyz, xyz1), nbytes=nb, extra=f"SA1 r={r} K={k}")
    for (r, k) in [(0.2, 64), (0.4, 128)]:
        nb = B * (12 * 512 + 12 * 128 + 4 * 128 * k)
        timed("ball_query", lambda: pn.ball_query(r, k, xyz1, xyz2), nbytes=nb, extra=f"SA2 r={r} K={k}")
    # multi-radius
    outs = [torch.zeros(B, 512, k, dtype=torch.int32, device=dev) for k in (32, 64, 128)]
    radii = (ctypes.c_float * 3)(0.05, 0.1, 0.2)
    ns = (ctypes.c_int * 3)(32, 64, 128)
    ptrs = (ctypes.c_void_p * 3)(*[o.data_ptr() for o in outs])
    nb = B * sum(12 * N + 12 * 512 + 4 * 512 * k for k in (32, 64, 128))
    timed("ball_query", lambda: _lib.call("captra_ball_query_multi", B, N, 512, 3, ctypes.cast(radii, ctypes.c_void_p),
                                           ctypes.cast(ns, ctypes.c_void_p), xyz1.data_ptr(), xyz.data_ptr(),
                                           ctypes.cast(ptrs, ctypes.c_void_p)), nbytes=nb, extra="SA1 3 radii fused")

    # ---- group (algorithmic bytes 4CN + 4MK + 4CMK per cloud)
    xyz_cn = xyz.transpose(1, 2).contiguous()
    for k in (32, 64, 128):
        idx = pn.ball_query({32: 0.05, 64: 0.1, 128: 0.2}[k], k, xyz, xyz1)
        nb = B * (4 * 3 * N + 4 * 512 * k + 4 * 3 * 512 * k)
        timed("group_points", lambda: pn.grouping_operation(xyz_cn, idx), nbytes=nb, extra=f"SA1 xyz C=3 K={k}")
    feat = torch.randn(B, 320, 512, device=dev)
    for (r, k) in [(0.2, 64), (0.4, 128)]:
        idx = pn.ball_query(r, k, xyz1, xyz2)
        nb = B * (4 * 320 * 512 + 4 * 128 * k + 4 * 320 * 128 * k)
        timed("group_points", lambda: pn.grouping_operation(feat, idx), nbytes=nb, extra=f"SA2 feat C=320 K={k}")

    # ---- three_nn / interpolate
    timed("three_nn", lambda: pn.three_nn(xyz, xyz1), extra="FP1 4096 vs 512")
    timed("three_nn", lambda: pn.three_nn(xyz1, xyz2), extra="FP2 512 vs 128")
    d, i = pn.three_nn(xyz, xyz1)
    w = torch.softmax(-d, -1).contiguous()
    f1 = torch.randn(B, 128, 512, device=dev)
    nb = B * (4 * 128 * 512 + 24 * N + 4 * 128 * N)
    timed("three_interpolate", lambda: pn.three_interpolate(f1, i, w), nbytes=nb, extra="FP1 C=128")

    # device copy reference for the achievable HBM rate
    a = torch.empty(256 * 1024 * 1024 // 4, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"device copy 256 MiB: {2 * a.numel() * 4 / ms / 1e6:.1f} GB/s (read+write)")


if __name__ == "__main__":
    main()
