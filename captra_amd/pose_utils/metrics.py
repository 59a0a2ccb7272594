"""Pose error metrics (mirrors the reference's pose_utils/metrics.py:5-45)."""
from __future__ import annotations

import math

import torch


def rot_diff_rad(rot1: torch.Tensor, rot2: torch.Tensor, yaxis_only: bool = False) -> torch.Tensor:
    if yaxis_only:
        cosv = (rot1[..., 1] * rot2[..., 1]).sum(dim=-1)
    else:
        m = torch.matmul(rot1, rot2.transpose(-1, -2))
        cosv = (m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2] - 1) / 2.0
    return torch.acos(torch.clamp(cosv, min=-1.0, max=1.0))


def rot_diff_degree(rot1, rot2, yaxis_only: bool = False):
    return rot_diff_rad(rot1, rot2, yaxis_only=yaxis_only) / math.pi * 180.0


def trans_diff(trans1: torch.Tensor, trans2: torch.Tensor) -> torch.Tensor:
    d = trans1 - trans2
    return torch.linalg.vector_norm(d.reshape(d.shape[:-1]), dim=-1)


def scale_diff(scale1: torch.Tensor, scale2: torch.Tensor) -> torch.Tensor:
    return torch.abs(scale1 - scale2)
