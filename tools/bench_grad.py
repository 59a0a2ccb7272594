import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from captra_amd import _lib
from captra_amd.pointnet_lib import pointnet2_utils as pn
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for name, B, C, N, M, K in [("SA2 feat", 12, 320, 512, 128, 128), ("SA2 feat K64", 12, 320, 512, 128, 64), ("SA1 xyz", 12, 3, 4096, 512, 128), ("SA1 feat", 12, 6, 4096, 512, 64)]:
    feat = torch.randn(B, C, N, generator=g).to(dev).requires_grad_()
    idx = torch.randint(0, N, (B, M, K), generator=g, dtype=torch.int32).to(dev)
    go = torch.randn(B, C, M, K, generator=g).to(dev)
    for _ in range(3):
        feat.grad = None
        pn.grouping_operation(feat, idx).backward(go)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10):
        feat.grad = None
        pn.grouping_operation(feat, idx).backward(go)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    out = {n: _lib.prof_read(n) for n in _lib.prof_names()}
    print(name, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in out.items() if v[1]}, "us per launch")
for name, B, C, N, M in [("FP1 interp", 12, 128, 4096, 512), ("FP2 interp", 12, 256, 512, 128)]:
    feat = torch.randn(B, C, M, generator=g).to(dev).requires_grad_()
    idx = torch.randint(0, M, (B, N, 3), generator=g, dtype=torch.int32).to(dev)
    w = torch.rand(B, N, 3, generator=g).to(dev)
    go = torch.randn(B, C, N, generator=g).to(dev)
    for _ in range(3):
        feat.grad = None
        pn.three_interpolate(feat, idx, w).backward(go)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10):
        feat.grad = None
        pn.three_interpolate(feat, idx, w).backward(go)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    out = {n: _lib.prof_read(n) for n in _lib.prof_names()}
    print(name, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in out.items() if v[1]}, "us per launch")
