"""On-the-fly ball crop + resample of a depth frame, on the device.

Real NOCS tracking (`--nocs_otf True` in every scripts/track/nocs/*.sh of the reference) re-crops every frame around the
pose predicted for the previous one: project the ball's bounding box into the image, back-project the depth pixels inside
it, keep the points within `radius` of the predicted centre, duplicate-pad / randomly thin to at most 5 x num_points,
furthest-point-sample num_points of them, and derive labels and ground-truth NOCS (reference
datasets/nocs_data/nocs_data_process.py:92-109, 121-163, 43-50, 227-236; nocs_utils.py:5-43; data_utils.py:138-157;
called from network/models/model.py:425-452).  The reference does this in numpy on the host with a GPU round trip for the
FPS; here the frame's depth and mask stay on the device, the arithmetic is torch float64 exactly as numpy's, the
furthest-point sampling is captra_fps_gather, and the host only sees three scalars (pixel / point counts).

Ordering contract (it decides which point FPS starts from): candidate pixels in row-major image order, ball members in
that order, duplication by whole-list doubling, the > 5 x num_points thinning by `numpy.random.permutation` (drawn on the
host from numpy's global generator, as the reference does, so that seeded runs agree).
"""
from __future__ import annotations

import numpy as np
import torch

NOCS_REAL_INTRINSICS = np.array([[591.0125, 0.0, 322.525], [0.0, 590.16775, 244.11084], [0.0, 0.0, 1.0]])   # nocs_data_process.py:20


def proj_corners(height: int, width: int, center, radius: float, intrinsics=NOCS_REAL_INTRINSICS) -> np.ndarray:
    """Image-space bounding box [[row_min, col_min], [row_max, col_max]] (inclusive, clamped) of the axis-aligned cube of
    half-size max(radius, 0.05) around `center` (camera frame, metres, looking down -z); nocs_data_process.py:136-148."""
    radius = max(float(radius), 0.05)
    c = np.asarray(center, np.float64).reshape(3)
    lo, hi = c - radius, c + radius
    box = np.array([[x, y, z] for y in (lo[1], hi[1]) for x in (lo[0], hi[0]) for z in (lo[2], hi[2])], np.float64) * 1000.0
    homog = -box / box[:, 2:3]
    homog[:, 2] = -homog[:, 2]
    uv = (np.asarray(intrinsics, np.float64) @ homog.T).T[:, :2].astype(np.int32)
    rows, cols = height - uv[:, 1], uv[:, 0]
    out = np.array([[rows.min(), cols.min()], [rows.max(), cols.max()]], np.int64)
    out[0] = np.maximum(out[0], 0)
    out[1] = np.minimum(out[1], np.array([height - 1, width - 1]))
    return out


def _device_fps(points_f32: torch.Tensor, num: int) -> torch.Tensor:
    from . import fused
    res = fused.fps_gather(points_f32.reshape(1, -1, 3).contiguous(), num)
    if res is None:   # beyond the register-resident kernel: the drop-in op
        from .pointnet_lib import pointnet2_utils as pn
        return pn.furthest_point_sample(points_f32.reshape(1, -1, 3).contiguous(), num).reshape(-1).long()
    return res[0].reshape(-1).long()


def crop_ball_from_depth(depth: torch.Tensor, mask: torch.Tensor, center, radius: float, num_points: int,
                         intrinsics=NOCS_REAL_INTRINSICS, fps_fn=_device_fps, _depth_retry: int = 0):
    """depth (H,W) integer millimetres, mask (H,W) bool (the tracked instance), both on the compute device
    -> (points (num_points,3) float64 camera frame, obj_mask (num_points,) bool).  nocs_data_process.py:92-109, 151-163."""
    H, W = depth.shape
    dev = depth.device
    box = proj_corners(H, W, center, radius, intrinsics)
    r0, c0, r1, c1 = int(box[0, 0]), int(box[0, 1]), int(box[1, 0]), int(box[1, 1])
    sub = depth[r0:r1 + 1, c0:c1 + 1]
    rc = torch.nonzero(sub > 0)                                   # row-major, like numpy.where
    rows, cols = rc[:, 0] + r0, rc[:, 1] + c0
    kinv = torch.from_numpy(np.linalg.inv(np.asarray(intrinsics, np.float64))).to(dev)
    grid = torch.stack([cols.double(), (H - rows).double(), torch.ones_like(cols, dtype=torch.float64)], dim=0)
    xyz = (kinv @ grid).t()                                        # (n,3) float64
    z = depth[rows, cols].float().double()
    pts = xyz * z[:, None] / xyz[:, 2:3]
    pts = torch.stack([pts[:, 0], pts[:, 1], -pts[:, 2]], dim=1) * 0.001
    raw_mask = mask[rows, cols].bool()

    c = torch.as_tensor(np.asarray(center, np.float64).reshape(1, 3), device=dev)
    dist = torch.sqrt(((pts - c) ** 2).sum(dim=-1))
    rad = max(float(radius), 0.05)
    idx = torch.empty(0, dtype=torch.long, device=dev)
    for _ in range(10):
        idx = torch.nonzero(dist <= rad).reshape(-1)
        if idx.numel() >= 10:
            break
        rad *= 1.10
    if idx.numel() == 0:
        idx = torch.arange(dist.numel(), device=dev)
    if idx.numel() == 0:
        if _depth_retry > 20:
            raise RuntimeError("crop_ball_from_depth: no valid depth pixel anywhere near the predicted centre")
        return crop_ball_from_depth(depth, mask, center, float(radius) * 1.2, num_points, intrinsics, fps_fn, _depth_retry + 1)
    while idx.numel() < num_points:
        idx = torch.cat([idx, idx])
    cand = pts[idx]
    if idx.numel() > 5 * num_points:                                # data_utils.py:146-152
        perm = torch.from_numpy(np.random.permutation(idx.numel())[:5 * num_points]).to(dev)
        picked = perm[fps_fn(cand[perm].float(), num_points)]
    else:
        picked = fps_fn(cand.float(), num_points)
    idx = idx[picked]
    return pts[idx], raw_mask[idx]


def full_data_from_depth(depth, mask, center, radius, gt_pose: dict, num_points: int, intrinsics=NOCS_REAL_INTRINSICS,
                         fps_fn=_device_fps) -> dict:
    """-> {'points' (N,3), 'labels' (N,) (0 = object, 1 = background), 'nocs' (N,3)} float64 / int64 device tensors;
    gt_pose {'rotation' (3,3), 'translation' (3,1), 'scale' ()} of the instance (nocs_data_process.py:43-50, 227-236)."""
    pts, obj = crop_ball_from_depth(depth, mask, center, radius, num_points, intrinsics, fps_fn)
    dev = pts.device
    rot = torch.as_tensor(np.asarray(gt_pose["rotation"], np.float64).reshape(3, 3), device=dev)
    trans = torch.as_tensor(np.asarray(gt_pose["translation"], np.float64).reshape(1, 3), device=dev)
    scale = float(np.asarray(gt_pose["scale"], np.float64).reshape(-1)[0])
    nocs = torch.zeros_like(pts)
    nocs[obj] = ((pts[obj] - trans) / scale) @ rot
    return {"points": pts, "labels": 1 - obj.long(), "nocs": nocs}
