"""Golden fixture G12-64: the gradients of G12's training step, from the reference's own `Trainer.update` run in FLOAT64.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train64.py [--ref /root/reference]

G12 (make_golden_train.py) records the reference's fp32 CPU step, whose backward sums in a thread-dependent order; a
gradient test against it cannot be tighter than the fp32 noise of BOTH sides.  Here the same step is run with every
floating tensor in double precision, so that the fixture is the real-arithmetic gradient of the reference's loss to ~1e-12
and the GPU's fp32 backward is measured against that alone.

What stays float32, on purpose: the DISCRETE choices the fp32 product path makes — farthest-point picks, ball-query
neighbour sets, the three nearest neighbours — are computed on float32 copies of the coordinates, exactly as in
make_golden_train.py, so that both runs differentiate the same piecewise-smooth function; and the random draws of the pose
noise / pair-wise-match samples come from the float32 streams (a float64 draw consumes the generator differently).

Run-time adjustments only (monkey patches, never edits): default dtype float64, `Tensor.float()` returns double, float32
random draws widened, plus the three patches of make_golden_train.py.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from tests import clouds  # noqa: E402
from tests.golden.make_golden import import_reference  # noqa: E402
from tests.golden.make_golden_train import CASES, TORCH_SEED, ForceFpsStartZero, probe_names, ref_cfg  # noqa: E402
from tests.weights import make_state_dict  # noqa: E402

F32 = torch.float32


def f32(t):
    return t.to(F32)


def cuda_semantics_f32_choices(pu):
    """Index selection on float32 copies (the product path's arithmetic), distances returned in the caller's precision."""

    def three_nn(a, b):
        a32, b32 = f32(a), f32(b)
        diff = a32[:, :, None, :] - b32[:, None, :, :]
        sq = diff * diff
        d2 = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        _, i = d2.sort(dim=-1, stable=True)
        i = i[:, :, :3]
        nb = torch.gather(b[:, None].expand(-1, a.shape[1], -1, -1), 2, i[..., None].expand(-1, -1, -1, 3))
        d = ((a[:, :, None, :] - nb) ** 2).sum(-1)
        return torch.sqrt(d), i

    def query_ball_point(radius, nsample, xyz, new_xyz):
        xyz, new_xyz = f32(xyz), f32(new_xyz)
        B, N, _ = xyz.shape
        r2 = torch.tensor(radius, dtype=F32) * torch.tensor(radius, dtype=F32)
        out = []
        for b in range(B):
            diff = new_xyz[b, :, None, :] - xyz[b, None, :, :]
            sq = diff * diff
            d2 = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
            cand = torch.where(d2 < r2, torch.arange(N).view(1, N), torch.full((1, 1), N))
            first_k = cand.sort(dim=-1)[0][:, :nsample]
            first = first_k[:, :1].clone()
            first[first == N] = 0
            out.append(torch.where(first_k == N, first.expand_as(first_k), first_k))
        return torch.stack(out)

    fps = pu.farthest_point_sample

    def farthest_point_sample(xyz, npoint):
        torch.set_default_dtype(F32)      # its running-distance buffer is created with the default dtype
        try:
            return fps(f32(xyz), npoint)
        finally:
            torch.set_default_dtype(torch.float64)

    pu.three_nn = three_nn
    pu.query_ball_point = query_ball_point
    pu.farthest_point_sample = farthest_point_sample


class Float64Everywhere:
    """default dtype float64; `.float()` widens instead of narrowing; float32 random streams, widened after the draw."""

    def __enter__(self):
        self.saved = (torch.Tensor.float, torch.randn, torch.rand, torch.randn_like, torch.rand_like)
        _, randn, rand, randn_like, rand_like = self.saved
        torch.set_default_dtype(torch.float64)
        torch.Tensor.float = lambda t, *a, **k: t.double()
        torch.randn = lambda *a, **k: randn(*a, **{**k, "dtype": F32}).double()
        torch.rand = lambda *a, **k: rand(*a, **{**k, "dtype": F32}).double()
        torch.randn_like = lambda t, **k: randn_like(t, **{**k, "dtype": F32}).double()
        torch.rand_like = lambda t, **k: rand_like(t, **{**k, "dtype": F32}).double()

    def __exit__(self, *exc):
        torch.Tensor.float, torch.randn, torch.rand, torch.randn_like, torch.rand_like = self.saved
        torch.set_default_dtype(F32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    pu = import_reference(args.ref)
    assert not pu.CUDA
    torch.set_num_threads(8)
    from trainer import Trainer
    g12 = np.load(HERE / "g12_train.npz")
    out = {}
    with Float64Everywhere():
        cuda_semantics_f32_choices(pu)
        for tag, ntype, config, cat, objcfg, kind, wseed in CASES:
            cfg = ref_cfg(config, cat, objcfg)
            with contextlib.redirect_stdout(io.StringIO()):
                trainer = Trainer(cfg)
            model = trainer.model
            shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
            model.load_state_dict(make_state_dict(shapes, seed=wseed))
            assert next(model.parameters()).dtype == torch.float64
            data = clouds.make_trajectory(kind, 2, 2, seed=3)[1]
            torch.manual_seed(TORCH_SEED)
            np.random.seed(TORCH_SEED)
            with ForceFpsStartZero():
                loss_dict = trainer.update(data)
            for k, v in loss_dict.items():
                out[f"{tag}/loss/{k}"] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v, np.float64)
            params = dict(model.named_parameters())
            assert all(p.grad is None or p.grad.dtype == torch.float64 for p in params.values())
            out[f"{tag}/grad_norm"] = np.float64(np.sqrt(sum(float((p.grad ** 2).sum()) for p in params.values() if p.grad is not None)))
            worst = 0.0
            for n in probe_names(params):
                g = params[n].grad.numpy().copy()
                out[f"{tag}/grad/{n}"] = g
                ref32 = g12[f"{tag}/grad/{n}"]
                worst = max(worst, float(np.abs(g - ref32).max() / np.abs(g).max()))
            dl = max(abs(float(out[f"{tag}/loss/{k}"]) - float(g12[f"{tag}/loss/{k}"])) for k in loss_dict if f"{tag}/loss/{k}" in g12.files and "loss" in k)
            print(f"{tag}: grad_norm {out[f'{tag}/grad_norm']:.9g} (fp32 run {float(g12[f'{tag}/grad_norm']):.9g}); max |loss64 - loss32| {dl:.3g}; "
                  f"fp32 reference run vs this: worst probe max|dg|/max|g| = {worst:.3g}", flush=True)
    np.savez_compressed(HERE / "g12_train64.npz", **out)
    print("wrote", HERE / "g12_train64.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
