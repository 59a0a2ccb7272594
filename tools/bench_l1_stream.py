"""Level-1 stream kernel (captra_sa1_stream_bf16) against the launches it replaces (captra_fps_gather + captra_ball_query_multi + 3 x
captra_sa_scale_bf16 per network), both networks of a rigid frame on the same clouds; us per call over 20 calls, graph-captured."""
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, '.')
from captra_amd import _lib, fused  # noqa: E402
from captra_amd import synthetic as clouds  # noqa: E402

dev = torch.device('cuda:0')
WIDTHS = ((32, 32, 64), (64, 64, 128), (64, 96, 128))
KS, RADII = (32, 64, 128), (0.05, 0.1, 0.2)


def module(cf, seed):
    rng = np.random.default_rng(seed)
    folded = []
    for chans in WIDTHS:
        dims = (cf + 3,) + chans
        folded.append([fused.pack(torch.from_numpy((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32)).to(dev),
                                  torch.from_numpy(0.1 * rng.standard_normal(dims[i + 1]).astype(np.float32)).to(dev)) for i in range(3)])
    return SimpleNamespace(training=False, knn=False, nsample_list=list(KS), radius_list=list(RADII), npoint=512, _folded=folded)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


fused.set_mlp_dtype("bf16")
mods = [module(0, 1), module(3, 2)]
one_net = "--one" in sys.argv
if one_net:
    sys.argv.remove("--one")
    mods = mods[:1]
batches = [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64]
for B in batches:
    x_n3 = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)]).astype(np.float32)).to(dev).contiguous()
    x_cn = x_n3.transpose(1, 2).contiguous()
    feats = [None, x_cn][:len(mods)]

    def sampler():
        return fused.fps_gather(x_n3, 512)

    def three():
        idx, n3, cn = fused.fps_gather(x_n3, 512)
        lists = fused.ball_query_multi(RADII, KS, x_n3, n3)
        for mod, feat in zip(mods, feats):
            out = torch.empty(B, 320, 512, device=dev)
            off = 0
            for layers, l in zip(mod._folded, lists):
                fused.sa_scale_bf16(feat, x_cn, n3, l, layers, out, off)
                off += layers[-1].cout

    planes = fused.bq_planes(x_n3)
    use_planes = [True]

    def stream():
        return fused.sa1_stream_bf16(x_n3, x_cn, mods, feats, planes=planes if use_planes[0] else None, m2=128)

    t_s, t_3 = timed(sampler), timed(three)
    line = f"B={B:3d}: sampler alone {t_s:7.1f} us, seven launches {t_3:7.1f} us"
    lib = _lib.lib()
    for whole, fine, dbg in ((0, 32, 0), (256, 32, 0), (0, 64, 0), (0, 0, 0), (4, 32, 0), (0, 32, 8), (0, 32, 16), (256, 32, 16)):  # (dbg & 1 reads lists nobody wrote: not in a sweep)
        lib.captra_sa1_stream_set_whole(whole)
        lib.captra_sa1_stream_set_fine(fine + (dbg << 16))
        print(f"  B={B} whole {whole} fine {fine} dbg {dbg} ...", end="", flush=True)
        t_ = timed(stream)
        got_ = stream()
        torch.cuda.synchronize()
        sp_ = fused.sa1_stream_spans(got_[5])
        print(f" {t_:6.1f} us (in-kernel: sampler {sp_[0]:6.1f}, all {sp_[1]:6.1f})", flush=True)
        line += f", [whole {whole} fine {fine} dbg {dbg}] {t_:6.1f}"
    lib.captra_sa1_stream_set_whole(0)
    lib.captra_sa1_stream_set_fine(32)
    use_planes[0] = False
    line += f", no plane image {timed(stream):7.1f} us"
    got = stream()
    torch.cuda.synchronize()
    print(line + f", gave up: {fused.sa1_stream_gave_up(got[5])}", flush=True)
