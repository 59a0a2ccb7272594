"""The three neck modules (SA3, FP3, FP2) in the bf16 mode: one launch each (captra_neck_chain_bf16) against the layer-by-layer route; us per
module call at the bench's batch sizes, graph-captured."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from captra_amd import fused  # noqa: E402
from captra_amd.pointnet_utils import PointNetFeaturePropagation, PointNetSetAbstraction  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)


def layers(dims):
    return [fused.pack(torch.from_numpy((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32)).to(dev),
                       torch.from_numpy(0.1 * rng.standard_normal(dims[i + 1]).astype(np.float32)).to(dev)) for i in range(len(dims) - 1)]


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
sa3 = PointNetSetAbstraction(None, None, None, 515, [256, 512, 1024], group_all=True).to(dev).eval()
sa3._folded = layers((515, 256, 512, 1024))
fp3 = PointNetFeaturePropagation(1536, [256, 256]).to(dev).eval()
fp3._folded = layers((1536, 256, 256))
fp2 = PointNetFeaturePropagation(576, [256, 128]).to(dev).eval()
fp2._folded = layers((576, 256, 128))
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 32]:
    x1 = torch.rand(B, 512, 3, device=dev) - 0.5
    x2 = x1[:, :128].contiguous()
    xyz1, xyz2 = x1.transpose(1, 2).contiguous(), x2.transpose(1, 2).contiguous()
    feat2 = torch.rand(B, 512, 128, device=dev)
    pooled = torch.rand(B, 1024, 1, device=dev)
    zero = torch.zeros(B, 3, 1, device=dev)
    f1 = torch.rand(B, 320, 512, device=dev)
    f3 = torch.rand(B, 256, 128, device=dev)
    nn = fused.three_nn_weights(x1, x2)
    jobs = {"SA3": lambda: sa3(xyz2, feat2), "FP3": lambda: fp3(xyz2, zero, feat2, pooled),
            "FP2": lambda: fp2(xyz1, xyz2, f1, f3, xyz1_n3=x1, xyz2_n3=x2, nn=nn)}
    line = f"B={B:3d}:"
    for name, fn in jobs.items():
        fused.USE_NECK_CHAIN = True
        t1 = timed(fn)
        fused.USE_NECK_CHAIN = False
        t0 = timed(fn)
        line += f"  {name} one launch {t1:6.1f} us / layer by layer {t0:6.1f} us"
    fused.USE_NECK_CHAIN = True
    from captra_amd import _lib
    for n in (1, 2, 4):
        _lib.lib().captra_neck_chain_set_split(n)
        line += f"  SA3 split {n}: {timed(jobs['SA3']):6.1f}"
    _lib.lib().captra_neck_chain_set_split(4)
    print(line, flush=True)
