"""Micro-benchmark of one fused SA scale launch (sa_fused.hip) at the bench workload's shapes.
Usage: python tools/bench_sa_fused.py [--which sa2s2|sa2s1|sa1s3|sa1s2|sa1s1|all] [--clouds 64] [--iters 10] [--layered]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import _lib, fused  # noqa: E402

SHAPES = {  # name: (cfeat, (c1,c2,c3), n, m, k)
    "sa1s1": (3, (32, 32, 64), 4096, 512, 32), "sa1s2": (3, (64, 64, 128), 4096, 512, 64),
    "sa1s3": (3, (64, 96, 128), 4096, 512, 128), "sa2s1": (320, (128, 128, 256), 512, 128, 64),
    "sa2s2": (320, (128, 196, 256), 512, 128, 128),
    "sa1s1x": (0, (32, 32, 64), 4096, 512, 32), "sa1s2x": (0, (64, 64, 128), 4096, 512, 64), "sa1s3x": (0, (64, 96, 128), 4096, 512, 128),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="all")
    ap.add_argument("--clouds", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layered", action="store_true")
    ap.add_argument("--wn", type=int, default=0)
    ap.add_argument("--pre", action="store_true", help="pre-transformed first layer (captra_sa_scale_pre) where supported; time includes the v1 launch")
    ap.add_argument("--pipe", action="store_true", help="pipelined SA2 kernel (captra_sa_scale_pre_pm); time includes the point-major v1 launch")
    ap.add_argument("--bf16", action="store_true", help="the bf16-native kernel (csrc/sa_bf16.hip); time includes the point-major v1 launch of the SA2 scales")
    ap.add_argument("--zeros", action="store_true", help="all-zero features / weights: same instruction stream at lower power (DVFS probe)")
    ap.add_argument("--mode", type=int, default=0, help="0 = register-resident kernels where instantiated, 1 = generic LDS kernel")
    ap.add_argument("--phases", action="store_true", help="debug: in-kernel s_memtime phase breakdown (needs a library built with CAPTRA_HIPCC_EXTRA=-DCAPTRA_SA_PROF=1)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import ctypes
    _lib.lib().captra_sa_fused_set_wn(ctypes.c_int(a.wn))
    B = a.clouds
    names = list(SHAPES) if a.which == "all" else [a.which]
    for name in names:
        cfeat, ch, n, m, k = SHAPES[name]
        g = torch.Generator(device="cpu").manual_seed(0)
        xyz = (torch.rand(B, 3, n, generator=g) - 0.5).to(dev)
        feat = torch.randn(B, cfeat, n, generator=g).to(dev) if cfeat else None
        new_xyz = (torch.rand(B, m, 3, generator=g) - 0.5).to(dev)
        idx = torch.randint(0, n, (B, m, k), generator=g, dtype=torch.int32).to(dev)
        dims = (cfeat + 3,) + ch
        layers = [fused.pack((torch.randn(dims[i], dims[i + 1], generator=g) / dims[i] ** 0.5).to(dev), torch.randn(dims[i + 1], generator=g).to(dev)) for i in range(3)]
        out = torch.empty(B, ch[2], m, device=dev)
        if a.zeros:
            xyz.zero_(); new_xyz.zero_()
            if feat is not None:
                feat.zero_()
            for lin in layers:
                lin.wt.zero_(); lin.bias.zero_()

        def run():
            if a.bf16:
                fused.sa_scale_bf16(feat, xyz, new_xyz, idx, layers, out, 0)
                return
            if a.layered:
                y = fused.sa_group_mlp(feat, xyz, new_xyz, idx, layers[0])
                y = fused.pointwise_mlp(y, layers[1], fused.ACT_RELU)
                fused.mlp_max(y, layers[2], out, 0)
            elif a.pipe and feat is not None and fused.sa_scale_pipe_supported(cfeat, layers, m, k):
                v1 = fused.sa_first_layer_pre_pm(feat, layers[0])
                fused.sa_scale_pre_pm(v1, xyz, new_xyz, idx, layers, out, 0, cfeat)
            elif a.pre and feat is not None and fused.sa_scale_pre_supported(cfeat, layers, k):
                v1 = fused.sa_first_layer_pre(feat, layers[0])
                fused.sa_scale_pre(v1, xyz, new_xyz, idx, layers, out, 0, cfeat)
            else:
                fused.sa_scale_fused(feat, xyz, new_xyz, idx, layers, out, 0)

        if a.bf16:
            fused.set_mlp_dtype("bf16")
            assert fused.sa_scale_bf16_supported(cfeat, layers, k), name
        if not a.layered and not a.bf16:
            _lib.lib().captra_sa_fused_set_mode(ctypes.c_int(1))
            fused.sa_scale_fused(feat, xyz, new_xyz, idx, layers, out, 0)
            ref = out.clone()
            _lib.lib().captra_sa_fused_set_mode(ctypes.c_int(a.mode))
            out.zero_()
            run()
            torch.cuda.synchronize()
            print(f"  {name}: mode {a.mode} vs generic LDS kernel: bit-exact = {torch.equal(ref, out)}  (max |diff| {float((ref - out).abs().max()):.3g})")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        if a.phases and not a.layered:
            cnt = torch.zeros(10, dtype=torch.int64, device=dev)
            _lib.lib().captra_sa_fused_set_prof(ctypes.c_void_p(cnt.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            print(f"  timed launch took {e0.elapsed_time(e1) * 1e3:.1f} us")
            _lib.lib().captra_sa_fused_set_prof(ctypes.c_void_p(0))
            c = cnt.tolist()
            waves = max(c[9], 1)
            labels = (["ids + bias staging + barrier", "L1 (v1 gather)", "L2", "L3", "maxima + store"] if a.bf16 else      # sa2_bf16_kernel: per pass
                      ["loop top", "L1", "L2", "L3", "store (per centre)"] if a.pipe else      # sa_wave_pipe_kernel: per tile
                      ["loop top", "L1", "L2 + next gather", "L3 + max", "store (per centre)"] if cfeat <= 3 else   # sa_wave_lds_kernel: per tile
                      ["start", "L1", "L2", "L3", "end-barrier"])                 # sa_wave_kernel's timers (streamed-weight scales)
            tot = sum(c[:len(labels)]) / waves
            if c[6]:
                print(f"  wave life: {c[5] / waves:.0f} shader cycles in {c[6] / waves / 100:.1f} us (s_memrealtime) -> shader clock {c[5] / c[6] * 0.1:.3f} GHz")
            print(f"  {name}: per-wave cycles, kernel {tot:.0f}: " + "  ".join(f"{l} {c[i] / waves:.0f}" for i, l in enumerate(labels)))
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        ms = sum(_lib.prof_read(nm)[0] for nm in _lib.prof_names()) / a.iters
        flops = 2.0 * B * m * k * sum(dims[i] * dims[i + 1] for i in range(3))
        if a.bf16 and cfeat > 3:
            flops = 2.0 * B * (m * k * (3 * dims[1] + dims[1] * dims[2] + dims[2] * dims[3]) + n * cfeat * dims[1])   # as executed
        print(f"{name:6s} {'layered' if a.layered else 'fused':7s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  ({flops / 1e9:.1f} GFLOP)", flush=True)


if __name__ == "__main__":
    main()
