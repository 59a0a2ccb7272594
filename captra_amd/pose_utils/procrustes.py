"""Masked / batched Procrustes pieces with every SVD solved ON DEVICE.

Mirrors the API of the reference's pose_utils/procrustes.py — `rotate_pts_batch` (l.25-56),
`rot_around_yaxis_to_3d` (l.69-75), `rotate_pts_mask` (l.110-114), `scale_pts_mask` (l.117-120),
`translate_pts_mask` (l.123-129), `transform_pts_mask` (l.132-164), `rotate_pts_2d_batch`
(l.167-204), `transform_pts_2d_mask` (l.213-228) — without the `.cpu()` / `torch.svd` /
`.to(device)` round trip of l.27-47 and l.170-189:
  * 3x3: HIP kernel captra_procrustes_rot3 (cross-covariance reduction + Jacobi in one launch);
  * 2x2: closed form — U diag(1, det(UV^T)) V^T of M is the rotation by atan2(M10-M01, M00+M11).
The fused whole-fit kernel used by the track loop is pose_fit.part_fit_st_no_ransac.
"""
from __future__ import annotations

import torch

from .. import _lib as L

EPS = 1e-6


def rotate_pts_batch(source: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """source, target (..., N, 3) -> R (..., 3, 3) = U diag(1,1,det(UV^T)) V^T, U S V^T = target^T source."""
    lead = source.shape[:-2]
    n = source.shape[-2]
    src = source.reshape(-1, n, 3).float().contiguous()
    tgt = target.expand_as(source).reshape(-1, n, 3).float().contiguous()
    L.require_device(src, tgt)
    rot = torch.empty(src.shape[0], 3, 3, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        L.call("captra_procrustes_rot3", src.shape[0], n, L.ptr(src), L.ptr(tgt), L.ptr(rot))
    return rot.reshape(lead + (3, 3))


def rot_around_yaxis_to_3d(rot_2d: torch.Tensor) -> torch.Tensor:
    xx, xz, zx, zz = rot_2d[..., 0, 0], rot_2d[..., 0, 1], rot_2d[..., 1, 0], rot_2d[..., 1, 1]
    one, zero = torch.ones_like(xx), torch.zeros_like(xx)
    return torch.stack([xx, zero, xz, zero, one, zero, zx, zero, zz], dim=-1).reshape(xx.shape + (3, 3))


def rotate_pts_mask(source, target, w):
    w = torch.sqrt(w + EPS)
    return rotate_pts_batch(source * w, target * w)


def scale_pts_mask(source, target, w):
    return torch.sum(source * target * w, dim=(-1, -2)) / (torch.sum(source * source * w, dim=(-1, -2)) + EPS)


def translate_pts_mask(source, target, w):
    """source, target (..., 3, N); w (..., N, 1) -> (..., 3, 1) weighted mean of target - source."""
    w = w.transpose(-1, -2)
    w_sum = torch.clamp(w.sum(dim=-1, keepdim=True), min=1.0)
    return torch.sum((target - source) * (w / w_sum), dim=-1, keepdim=True)


def rotate_pts_2d_batch(source, target):
    """source, target (..., N, 2) already centred -> (..., 2, 2) in-plane rotation (identity when
    the cross-covariance is degenerate, as the reference's validity fallback l.197-204)."""
    m = torch.matmul(target.transpose(-1, -2), source).detach()
    a = m[..., 0, 0] + m[..., 1, 1]
    c = m[..., 1, 0] - m[..., 0, 1]
    h = torch.sqrt(a * a + c * c)
    ok = h > 0
    safe = torch.where(ok, h, torch.ones_like(h))
    cs = torch.where(ok, a / safe, torch.ones_like(h))
    sn = torch.where(ok, c / safe, torch.zeros_like(h))
    return torch.stack([cs, -sn, sn, cs], dim=-1).reshape(cs.shape + (2, 2))


def rotate_pts_2d_mask(source, target, w):
    return rotate_pts_2d_batch(source * w, target * w)


def _masked_center(x, mask):
    return torch.sum(x * mask, dim=-2, keepdim=True) / torch.clamp(torch.sum(mask, dim=-2, keepdim=True), min=1.0)


def transform_pts_2d_mask(source, target, mask):
    """source, target (B,P,N,2), mask (B,P,N,1) -> (rotation (B,P,2,2), translation (B,P,2,1))."""
    sc = (source - _masked_center(source, mask)) * mask
    tc = (target - _masked_center(target, mask)) * mask
    rotation = rotate_pts_2d_mask(sc, tc, mask)
    translation = translate_pts_mask(torch.matmul(rotation, source.transpose(-1, -2)), target.transpose(-1, -2), mask)
    return rotation, translation


def transform_pts_mask(source, target, mask, weights, given_scale=None, rotation=None, sym=False):
    """source, target (..., N, 3); mask, weights (..., N, 1); rotation (..., 3, 3) or None.
    Returns (rotation, scale, translation) with the semantics of procrustes.py:132-164."""
    sc = (source - _masked_center(source, mask)) * mask
    tc = (target - _masked_center(target, mask)) * mask
    if rotation is None:
        rotation = rotate_pts_mask(sc, tc, weights)
    if sym:
        canon_target = torch.matmul(target, rotation)
        rot_2d, _ = transform_pts_2d_mask(source[..., [0, 2]], canon_target[..., [0, 2]], weights)
        rotation = torch.matmul(rotation, rot_around_yaxis_to_3d(rot_2d))
    if given_scale is not None:
        scale = given_scale
    else:
        scale = scale_pts_mask(torch.matmul(sc, rotation.transpose(-1, -2)), tc, weights)
    translation = translate_pts_mask(scale.reshape(scale.shape + (1, 1)) * torch.matmul(rotation, source.transpose(-1, -2)),
                                     target.transpose(-1, -2), weights)
    return rotation, scale, translation


# ---- unmasked (every point counts) variants of the same API (reference l.59-66, 78-107, 231-243) ----------------------------
def scale_pts_batch(source, target):
    """(..., N, 3) x2 -> (...): least-squares scale of source onto target."""
    return torch.sum(source * target, dim=(-1, -2)) / (torch.sum(source * source, dim=(-1, -2)) + EPS)


def translate_pts_batch(source, target):
    """(..., 3, N) x2 -> (..., 3, 1): mean of target - source."""
    return torch.mean(target - source, dim=-1, keepdim=True)


def transform_pts_2d_batch(source, target):
    """(..., N, 2) x2 -> (in-plane rotation (..., 2, 2), translation (..., 2, 1)); (None, None) never occurs here: the
    closed-form 2x2 solve has no failure path (degenerate inputs give the identity)."""
    rotation = rotate_pts_2d_batch(source - source.mean(-2, keepdim=True), target - target.mean(-2, keepdim=True))
    return rotation, translate_pts_batch(torch.matmul(rotation, source.transpose(-1, -2)), target.transpose(-1, -2))


def transform_pts_batch(source, target, given_scale=None, rotation=None, sym=False):
    """source, target (..., N, 3) -> (R (..., 3, 3), s (...), t (..., 3, 1)) with target ~ s R source + t; `rotation` given
    skips the 3x3 Procrustes; `sym` refines it by an in-plane rotation about y fitted on the (x, z) coordinates."""
    src_c = source - source.mean(-2, keepdim=True)
    tgt_c = target - target.mean(-2, keepdim=True)
    if rotation is None:
        rotation = rotate_pts_batch(src_c, tgt_c)
    if sym:
        canon_target = torch.matmul(target, rotation)
        rot_2d, _ = transform_pts_2d_batch(source[..., [0, 2]], canon_target[..., [0, 2]])
        rotation = torch.matmul(rotation, rot_around_yaxis_to_3d(rot_2d))
    scale = given_scale if given_scale is not None else scale_pts_batch(torch.matmul(src_c, rotation.transpose(-1, -2)), tgt_c)
    translation = translate_pts_batch(scale[..., None, None] * torch.matmul(rotation, source.transpose(-1, -2)),
                                      target.transpose(-1, -2))
    return rotation, scale, translation
