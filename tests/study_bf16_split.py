"""Numerical study (not a test): how far do bf16-split shared-MLP products (x3 = hi*hi + hi*lo + lo*hi, x4 = + lo*lo,
x6 = three-way split) move the free-running G9p trajectories away from the reference's poses, next to the exact fp32 chain?
Answers whether an opt-in `bf16x3` mode could hold the 1e-4 golden contract.  CPU only, emulation in
numpy: operands rounded to bf16 (round-to-nearest-even), products accumulated in fp32.
    python tests/study_bf16_split.py [tag ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import model as OM  # noqa: E402
from oracle import ops as O  # noqa: E402
from tests.test_oracle_golden import G, _physical_setup  # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(x, parts):
    out, rest = [], x.astype(np.float32)
    for _ in range(parts):
        h = bf16(rest)
        out.append(h)
        rest = rest - h
    return out


def make_mlp(mode):
    terms = {"x1": [(0, 0)], "x3": [(0, 0), (0, 1), (1, 0)], "x4": [(0, 0), (0, 1), (1, 0), (1, 1)],
             "x6": [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}[mode]
    parts = 1 + max(max(t) for t in terms)

    def pointwise_mlp(x, wt, b, act):
        shape = x.shape
        x2 = x.reshape(shape[0], shape[1], -1)
        ws, xs = split(wt, parts), split(x2, parts)
        y = np.zeros((shape[0], wt.shape[1], x2.shape[2]), np.float32)
        for i, j in reversed(terms):                    # small terms first
            y += np.einsum("kc,bkl->bcl", ws[i], xs[j], optimize=True).astype(np.float32)
        y += b[None, :, None]
        if act == 1:
            y = np.maximum(y, 0)
        return y.reshape((shape[0], wt.shape[1]) + tuple(shape[2:]))
    return pointwise_mlp


def main():
    tags = sys.argv[1:] or ["bottle", "laptop"]
    g = np.load(G / "g9p_track.npz")
    exact = O.pointwise_mlp
    for tag in tags:
        cfg, sd, data, init = _physical_setup(tag)
        for mode in ("exact", "x6", "x4", "x3", "x1"):
            O.pointwise_mlp = exact if mode == "exact" else make_mlp(mode)
            try:
                poses, _ = OM.track(sd, cfg, data, init, "exact")
            finally:
                O.pointwise_mlp = exact
            worst = {k: max(float(np.abs(poses[i][k] - g[f"{tag}_{i}_{k}"]).max()) for i in range(1, len(data))) for k in ("rotation", "scale", "translation")}
            print(f"{tag:8s} {mode:5s} max |pose - reference golden| over all frames: " + "  ".join(f"{k} {v:.2e}" for k, v in worst.items()), flush=True)


if __name__ == "__main__":
    main()
