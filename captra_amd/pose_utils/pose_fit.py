"""Per-part scale/translation fit from NOCS <-> camera correspondences, one HIP launch.

Mirrors `part_fit_st_no_ransac` / `filter_model_valid` of the reference's pose_utils/pose_fit.py
(l.26-53).  The reference builds a one-hot mask and runs transform_pts_mask (~30 ATen kernels and
a host SVD for symmetric objects); here the labels go straight to captra_part_fit_st.
"""
from __future__ import annotations

import torch

from .. import _lib as L


def filter_model_valid(model: dict, valid: torch.Tensor) -> torch.Tensor:
    for key in ("scale", "translation", "rotation"):
        tmp = model[key] if key == "scale" else model[key].sum((-1, -2))
        valid = torch.logical_and(valid, torch.isfinite(tmp))
    return valid


def part_fit_st_cn(labels_i32, src_cn, tgt_cn, rotation, sym: bool, given_scale=None, tgt_per_part=False):
    """Channel-major fast path used by the track loop (no transposes):
    labels (B,N) int32, src_cn (B,P,3,N), tgt_cn (B,3,N) [or (B,P,3,N)], rotation (B,P,3,3)
    -> scale (B,P), translation (B,P,3,1), valid (B,P) bool."""
    L.require_device(labels_i32, src_cn, tgt_cn, rotation, given_scale)
    B, P, _, N = src_cn.shape
    dev = src_cn.device
    scale = torch.empty(B, P, dtype=torch.float32, device=dev)
    trans = torch.empty(B, P, 3, dtype=torch.float32, device=dev)
    valid = torch.empty(B, P, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.call("captra_part_fit_st", B, P, N, 1 if sym else 0, L.ptr(labels_i32), L.ptr(src_cn), L.ptr(tgt_cn),
               1 if tgt_per_part else 0, L.ptr(rotation), L.ptr(given_scale), L.ptr(scale), L.ptr(trans), L.ptr(valid))
    return scale, trans.unsqueeze(-1), valid.bool()


def part_fit_st_track(labels_i32, src_cn, pts_cn, pts_mean, rotation, prev_scale, prev_trans, sym: bool):
    """The track loop's fit in one launch (networks.py:219-232): target = pts (B,3,N) + pts_mean (B,3,1) formed inside the
    kernel, invalid fits keep prev_scale (B,P) / prev_trans (B,P,3,1) -> scale (B,P), translation (B,P,3,1), valid (B,P) bool."""
    B, P, _, N = src_cn.shape
    dev = src_cn.device
    pts_mean = pts_mean.reshape(B, 3).float().contiguous()
    prev_scale = prev_scale.float().contiguous()
    prev_trans = prev_trans.reshape(B, P, 3).float().contiguous()
    L.require_device(labels_i32, src_cn, pts_cn, pts_mean, rotation, prev_scale, prev_trans)
    scale = torch.empty(B, P, dtype=torch.float32, device=dev)
    trans = torch.empty(B, P, 3, dtype=torch.float32, device=dev)
    valid = torch.empty(B, P, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.call("captra_part_fit_st_track", B, P, N, 1 if sym else 0, L.ptr(labels_i32), L.ptr(src_cn), L.ptr(pts_cn), L.ptr(pts_mean),
               L.ptr(rotation), L.ptr(prev_scale), L.ptr(prev_trans), L.ptr(scale), L.ptr(trans), L.ptr(valid))
    return scale, trans.unsqueeze(-1), valid.bool()


def part_fit_st_no_ransac(labels, source, target, rotation, cfg, given_scale=None):
    """labels (B,N); source, target (B,P,N,3); rotation (B,P,3,3); cfg {'num_parts','sym'}
    -> ({'rotation','scale' (B,P),'translation' (B,P,3,1)}, valid (B,P) bool)."""
    src_cn = source.transpose(-1, -2).float().contiguous()
    tgt_cn = target.transpose(-1, -2).float().contiguous()
    rot = rotation.float().contiguous()
    gs = None if given_scale is None else given_scale.float().contiguous()
    scale, translation, valid = part_fit_st_cn(labels.int().contiguous(), src_cn, tgt_cn, rot, bool(cfg["sym"]),
                                               given_scale=gs, tgt_per_part=True)
    model = {"rotation": rotation, "scale": scale, "translation": translation}
    return model, filter_model_valid(model, valid)
