"""Per-layer error of the exact fp32 chain and of the f32x6 arithmetic against a float64 product (GPU): python tools/x6_error.py"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import fused
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for cin, cout, L in ((128, 512, 4096), (512, 512, 4096)):
    x = np.abs(rng.standard_normal((1, cin, L))).astype(np.float32)      # positive (post-ReLU-like) inputs
    wt = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lin = fused.pack(torch.from_numpy(wt).to(dev), torch.from_numpy(b).to(dev))
    xd = torch.from_numpy(x).to(dev)
    truth = np.einsum("kc,bkl->bcl", wt.astype(np.float64), x.astype(np.float64)) + b[None, :, None].astype(np.float64)
    scale = np.abs(truth).max()
    for mode in ("fp32", "f32x6"):
        with fused.use_mlp_dtype(mode):
            y = fused.pointwise_mlp_gn(xd, lin, None, fused.ACT_NONE).cpu().numpy().astype(np.float64)
        e = y - truth
        ulp = np.spacing(np.abs(truth).astype(np.float32)).astype(np.float64)
        print(f"{cin}->{cout} {mode:6s} max|e|/max|y| {np.abs(e).max() / scale:.2e}  rms(e/ulp) {np.sqrt(np.mean((e / ulp) ** 2)):.2f}  "
              f"mean(sign(y) e/ulp) {np.mean(np.sign(truth) * e / ulp):+.2f}", flush=True)
