"""Part pose containers: compose, perturb, evaluate.

Mirrors the used subset of the reference's pose_utils/part_dof_utils.py:
`part_model_batch_to_part` (l.70-75), `add_noise_to_part_dof` (l.78-98),
`merge_reenact_canon_part_pose` (l.124-134), `convert_pred_rtvec_to_matrix` (l.137-141),
`eval_part_full` (l.54-67, the 5°5cm accuracy gate), and for the training step `pose_with_part` (l.101-117),
`compute_parts_delta_pose` (l.144-159).
A part pose is a dict {'rotation' (B,P,3,3), 'translation' (B,P,3,1), 'scale' (B,P)}.
"""
from __future__ import annotations

import torch

from .metrics import rot_diff_degree, scale_diff, trans_diff
from .rotations import (compute_rotation_matrix_from_3d, compute_rotation_matrix_from_matrix,
                        noisy_rot_matrix)


def part_model_batch_to_part(part, num_parts: int, device):
    """[{'scale' (B,), 'translation' (B,3,1), 'rotation' (B,3,3)}]*P -> stacked over a part axis."""
    dim = part[0]["translation"].dim() - 2
    return {key: torch.stack([torch.as_tensor(part[p][key]) for p in range(num_parts)], dim=dim).float().to(device)
            for key in part[0].keys()}


def _cpu_random_like(base: torch.Tensor, rand_type: str) -> torch.Tensor:
    """Noise comes from the CPU generator (seed-stable across devices), then moves to base.device."""
    if rand_type == "uniform":
        r = torch.rand(base.shape) * 2.0 - 1.0
    elif rand_type == "normal":
        r = torch.randn(base.shape)
    else:
        raise ValueError(rand_type)
    return r.to(base.device)


def add_noise_to_part_dof(part: dict, cfg: dict) -> dict:
    """cfg: {'type', 'rotation' [rad], 'scale', 'translation'}.  Draw order (rotation angle,
    jitter quaternion, scale, translation norm, translation direction) is the reference's."""
    rand_type = cfg["type"]
    out = {"rotation": noisy_rot_matrix(part["rotation"], cfg["rotation"], type=rand_type).reshape(part["rotation"].shape)}
    out["scale"] = part["scale"] + _cpu_random_like(part["scale"], rand_type) * cfg["scale"]
    norm = _cpu_random_like(part["scale"], rand_type) * cfg["translation"]           # (B,P)
    direction = _cpu_random_like(part["translation"].squeeze(-1), rand_type)          # (B,P,3)
    direction = direction / torch.clamp(direction.norm(dim=-1, keepdim=True), min=1e-9)
    out["translation"] = part["translation"] + (direction * norm.unsqueeze(-1)).unsqueeze(-1)
    return out


def consume_noise_draws(part: dict, cfg: dict) -> None:
    """The random draws of add_noise_to_part_dof on a pose of this shape, and nothing else: the track loop's reference draws a
    perturbed pose every frame and never uses it (model.py:414-420); seeded runs must consume the generator identically,
    but the pose algebra and its five host-to-device copies per frame need not happen."""
    rand = torch.randn if cfg["type"] == "normal" else torch.rand
    if cfg["type"] not in ("normal", "uniform"):
        raise ValueError(cfg["type"])
    bp = tuple(part["scale"].shape)
    rand(bp)                # rotation angle            (noisy_rot_matrix)
    torch.randn(bp + (4,))  # jitter quaternion         (generate_random_quaternion: always normal)
    rand(bp)                # scale
    rand(bp)                # translation norm
    rand(bp + (3,))         # translation direction


def merge_reenact_canon_part_pose(part_dof: dict, delta: dict) -> dict:
    """Apply a canonical-frame delta to a pose: R = R_prev ΔR (and s, t when present)."""
    pose = {k: v.clone() for k, v in part_dof.items()}
    if "rotation" in delta:
        pose["rotation"] = torch.matmul(part_dof["rotation"], delta["rotation"])
    if "scale" in delta:
        pose["scale"] = delta["scale"].squeeze(-1) * part_dof["scale"]
    if "trans" in delta:
        pose["translation"] = part_dof["translation"] + part_dof["scale"][..., None, None] * torch.matmul(
            part_dof["rotation"], delta["trans"].unsqueeze(-1))
    return pose


def pose_with_part(model: dict, src: torch.Tensor) -> torch.Tensor:
    """Canonical points (B,P,K,3) posed by each part's (s, R, t): s * (src R^T) + t^T."""
    est = torch.matmul(src, model["rotation"].transpose(-1, -2)) * model["scale"][..., None, None]
    return est + model["translation"].transpose(-1, -2)


def compute_parts_delta_pose(init: dict, final: dict, canon: dict) -> dict:
    """The pose update that takes `init` to `final`, expressed in the frame of `canon` (the supervision target of the
    per-point rotation loss): init / final (B,P,…), canon (B,…) or (B,P,…)."""
    if canon["scale"].dim() < final["scale"].dim():
        canon = {k: v.unsqueeze(1) for k, v in canon.items()}
    s0, sf, sc = init["scale"], final["scale"], canon["scale"]
    t0, tf, tc = init["translation"], final["translation"], canon["translation"]
    r0, rf, rc = init["rotation"], final["rotation"], canon["rotation"]
    s_delta = sf / s0
    r_delta = torch.matmul(torch.matmul(rc.transpose(-1, -2), rf), torch.matmul(r0.transpose(-1, -2), rc))
    t = tf - tc
    if (t0 - tc).max() > 1e-7:
        t = t - s_delta[..., None, None] * torch.matmul(torch.matmul(rf, r0.transpose(-1, -2)), t0 - tc)
    t_delta = torch.matmul(rc.transpose(-1, -2), t) / sc[..., None, None]
    return {"scale": s_delta, "rotation": r_delta, "translation": t_delta}


def convert_pred_rtvec_to_matrix(pred: torch.Tensor, sym: bool) -> torch.Tensor:
    """(…, D) -> (…, 3, 3): D=3 y-axis for symmetric objects, D=9 matrix to re-orthogonalise."""
    if sym:
        return compute_rotation_matrix_from_3d(pred.reshape(-1, pred.shape[-1])).reshape(pred.shape[:-1] + (3, 3))
    return compute_rotation_matrix_from_matrix(pred.reshape(-1, 3, 3)).reshape(pred.shape[:-1] + (3, 3))


def eval_part_model(gt: dict, pred: dict, yaxis_only: bool = False) -> dict:
    return {"sdiff": scale_diff(gt["scale"], pred["scale"]),
            "tdiff": trans_diff(gt["translation"], pred["translation"]),
            "rdiff": rot_diff_degree(gt["rotation"], pred["rotation"], yaxis_only=yaxis_only)}


def eval_part_full(gt: dict, pred: dict, per_instance: bool = False, yaxis_only: bool = False):
    """Per-part errors + 5°5cm / 10°10cm hit rates; returns (batch means, per-instance values)."""
    d = eval_part_model(gt, pred, yaxis_only=yaxis_only)
    d["5deg5cm"] = torch.logical_and(d["rdiff"] <= 5.0, d["tdiff"] <= 0.05).float()
    d["10deg10cm"] = torch.logical_and(d["rdiff"] <= 10.0, d["tdiff"] <= 0.10).float()
    flat = {f"{k}_{i}": v[..., i] for k, v in d.items() for i in range(v.shape[-1])}
    per = {k: v.clone() for k, v in flat.items()} if per_instance else {}
    return {k: v.mean(dim=0) for k, v in flat.items()}, per
