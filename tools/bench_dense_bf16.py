"""The bf16-native dense layers of the rotation heads, one kernel at a time (HIP events, L2-cold rotation of buffers).
    python tools/bench_dense_bf16.py [--clouds 32] [--points 4096]
Prints us and TFLOP/s per variant (plain / ReLU / GroupNorm on load / statistics epilogue)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import fused  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L = a.clouds, a.points
    for cin, cout in [(128, 512), (512, 512), (512, 256)]:
        lin = fused.pack(torch.randn(cin, cout, device=dev) / cin ** 0.5, torch.randn(cout, device=dev))
        xs = [torch.randn(B, L, fused.pm_channels(cin), device=dev).to(torch.bfloat16) for _ in range(3)]
        ab = torch.randn(B, cin, 2, device=dev)
        if True:
            for name, kw in [("plain", {}), ("relu", {"act": fused.ACT_RELU}), ("gn-on-load", {"ab": ab}), ("stats", {"with_stats": True}),
                             ("gn-on-load+stats", {"ab": ab, "with_stats": True})]:
                for i in range(3):
                    fused.pointwise_mlp_bf16pm(xs[i % 3], lin, L, in_pm=True, out_pm=True, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    fused.pointwise_mlp_bf16pm(xs[i % 3], lin, L, in_pm=True, out_pm=True, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / a.iters
                print(f"{cin:4d} -> {cout:4d}  {name:18s} {us:8.1f} us  {2.0 * B * L * cin * cout / us / 1e6:7.1f} TFLOP/s", flush=True)

    # round 4: the LDS-tiled kernels (csrc/tile_bf16.hip) and the fused head
    def timeit(fn):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / a.iters

    for cin, cout in [(128, 512), (512, 512), (512, 256), (256, 128), (512, 1024)]:
        lin = fused.pack(torch.randn(cin, cout, device=dev) / cin ** 0.5, torch.randn(cout, device=dev))
        xs = [torch.randn(B, L, fused.pm_channels(cin), device=dev).to(torch.bfloat16) for _ in range(3)]
        ab = torch.randn(B, cin, 2, device=dev)
        for name, kw in [("tile plain", {}), ("tile gn-on-load", {"ab": ab}), ("tile stats", {"with_stats": True}), ("tile gn-on-load+stats", {"ab": ab, "with_stats": True})]:
            us = timeit(lambda i: fused.dense_bf16_tile(xs[i % 3], lin, **kw))
            print(f"{cin:4d} -> {cout:4d}  {name:22s} {us:8.1f} us  {2.0 * B * L * cin * cout / us / 1e6:7.1f} TFLOP/s", flush=True)
    lin1 = fused.pack(torch.randn(128, 512, device=dev) / 128 ** 0.5, torch.randn(512, device=dev))
    lin2 = fused.pack(torch.randn(512, 512, device=dev) / 512 ** 0.5, torch.randn(512, device=dev))
    xs = [torch.randn(B, L, 128, device=dev).to(torch.bfloat16) for _ in range(3)]
    ab1 = torch.randn(B, 512, 2, device=dev)
    us = timeit(lambda i: fused.head12_bf16_stats(xs[i % 3], lin1))
    print(f"head12 statistics pass (128 -> 512, nothing stored) {us:8.1f} us  {2.0 * B * L * 128 * 512 / us / 1e6:7.1f} TFLOP/s", flush=True)
    us = timeit(lambda i: fused.head12_bf16(xs[i % 3], lin1, ab1, lin2))
    print(f"head12 fused (128 -> 512 -> 512, y2 + stats)        {us:8.1f} us  {2.0 * B * L * (128 * 512 + 512 * 512) / us / 1e6:7.1f} TFLOP/s", flush=True)
    from captra_amd import _lib
    import ctypes
    if hasattr(_lib.lib(), "captra_tile_bf16_set_persistent"):
        for v in (0, 1, 0, 1):
            _lib.lib().captra_tile_bf16_set_persistent(ctypes.c_int(v))
            us = timeit(lambda i: fused.head12_bf16(xs[i % 3], lin1, ab1, lin2))
            print(f"   fused pair, persistent={v}: {us:8.1f} us", flush=True)
    if hasattr(_lib.lib(), "captra_tile_bf16_set_debug"):
        _lib.lib().captra_tile_bf16_set_persistent(ctypes.c_int(0))      # (the ablation switches are the one-tile-per-workgroup kernel's)
        for dbg in (1, 2, 4, 8, 16, 6, 14, 30):
            _lib.lib().captra_tile_bf16_set_debug(dbg)
            us = timeit(lambda i: fused.head12_bf16(xs[i % 3], lin1, ab1, lin2))
            print(f"   ablation dbg={dbg:2d} (1 A same frag, 2 no stores, 4 no stats, 8 no B reads, 16 no A loads): {us:8.1f} us", flush=True)
        _lib.lib().captra_tile_bf16_set_debug(0)
        _lib.lib().captra_tile_bf16_set_persistent(ctypes.c_int(1))


if __name__ == "__main__":
    main()
