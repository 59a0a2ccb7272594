"""Bounding-box IoU for the evaluation tables (reference pose_utils/bbox_utils.py).

Host-side numpy on a handful of boxes per frame -- evaluation, not the hot path.  Conventions of the reference:
box corners are ordered i -> (x = (i % 4) // 2, y = i // 4, z = i % 2) picking min/max per axis (l.64-72); rigid
(NOCS) categories score the axis-aligned extent of the posed corners (`nocs_iou_3d`, l.45-61); articulated parts
score oriented boxes by occupancy on a 50^3 grid spanning both boxes (`iou_3d`, l.28-42); symmetric objects take
the best of 20 rotations of the ground truth about its y axis (l.164-176).
"""
from __future__ import annotations

import numpy as np


def bbox_from_corners(corners) -> np.ndarray:
    """(..., 2, 3) [min; max] -> (..., 8, 3) corner points in the reference's order."""
    c = np.asarray(corners, dtype=np.float32)
    out = np.zeros(c.shape[:-2] + (8, 3), np.float32)
    for i in range(8):
        out[..., i, 0] = c[..., (i % 4) // 2, 0]
        out[..., i, 1] = c[..., i // 4, 1]
        out[..., i, 2] = c[..., i % 2, 2]
    return out


def pose_box(pose: dict, box: np.ndarray) -> np.ndarray:
    """s R p + t for every corner: pose {'rotation' (...,3,3), 'translation' (...,3,1), 'scale' (...)}, box (...,8,3)."""
    rot = np.asarray(pose["rotation"], np.float32)
    posed = np.matmul(box, np.swapaxes(rot, -1, -2)) * np.asarray(pose["scale"], np.float32)[..., None, None]
    return posed + np.swapaxes(np.asarray(pose["translation"], np.float32), -1, -2)


def pts_inside_box(pts: np.ndarray, box: np.ndarray) -> np.ndarray:
    """pts (...,3) "inside" the box (8,3) by the reference's test (l.11-25): 0 < (p - c4).u < u.u for u = c5 - c4,
    c7 - c4, c0 - c4.  With the corner order of bbox_from_corners, c7 - c4 is a face diagonal, not an edge -- the
    reference applies the test to that order all the same (l.95-103 -> l.28-42), and so does this restatement: the
    numbers are the reference's evaluation protocol, not an exact oriented-box volume ratio."""
    o = box[4]
    inside = np.ones(pts.shape[:-1], bool)
    for far in (5, 7, 0):
        u = box[far] - o
        proj = (pts - o) @ u
        inside &= (proj > 0) & (proj < float(u @ u))
    return inside


def iou_3d(box1: np.ndarray, box2: np.ndarray, nres: int = 50) -> float:
    """Occupancy IoU of two oriented boxes on an nres^3 grid over their joint extent (1 when both are empty)."""
    both = np.concatenate([box1, box2], 0)
    lo, hi = both.min(0), both.max(0)
    axes = [np.linspace(lo[d], hi[d], nres) for d in range(3)]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)
    in1, in2 = pts_inside_box(grid, box1), pts_inside_box(grid, box2)
    union = np.logical_or(in1, in2).sum()
    return 1.0 if union == 0 else float(np.logical_and(in1, in2).sum() / float(union))


def nocs_iou_3d(box1: np.ndarray, box2: np.ndarray) -> float:
    """IoU of the axis-aligned extents of two corner sets (the NOCS benchmark's protocol)."""
    lo1, hi1, lo2, hi2 = box1.min(0), box1.max(0), box2.min(0), box2.max(0)
    lo, hi = np.maximum(lo1, lo2), np.minimum(hi1, hi2)
    inter = 0.0 if np.min(hi - lo) < 0 else float(np.prod(hi - lo))
    union = float(np.prod(hi1 - lo1) + np.prod(hi2 - lo2) - inter)
    return inter / union


def _y_rotation(theta: float) -> np.ndarray:
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)


def part_iou(gt_boxes: list, pred_box: np.ndarray, nocs: bool) -> np.ndarray:
    """Per part, the best IoU of the predicted box against any of the candidate ground-truth boxes: (P,8,3) each -> (P,)."""
    fn = nocs_iou_3d if nocs else iou_3d
    return np.array([max(fn(g[p], pred_box[p]) for g in gt_boxes) for p in range(pred_box.shape[0])], np.float64)


def eval_single_part_iou(gt_corners, pred_corners, gt_pose: dict, pred_pose: dict, nocs: bool = False, sym: bool = False) -> dict:
    """One frame of one instance: corners (P,2,3), poses {'rotation' (P,3,3), 'translation' (P,3,1), 'scale' (P,)}
    -> {'npcs_iou', 'iou', 'gt_bbox_iou'}: (P,) each -- canonical-space boxes, posed predicted box vs posed ground truth,
    and the ground-truth box under the predicted pose vs under the ground-truth pose (reference l.160-198)."""
    gt_box, pred_box = bbox_from_corners(gt_corners), bbox_from_corners(pred_corners)
    if sym:
        n = 20
        gt_poses = [{"rotation": np.matmul(np.asarray(gt_pose["rotation"], np.float32), _y_rotation(2 * np.pi * i / n)),
                     "translation": gt_pose["translation"], "scale": gt_pose["scale"]} for i in range(n)]
    else:
        gt_poses = [gt_pose]
    gt_posed = [pose_box(p, gt_box) for p in gt_poses]
    return {"npcs_iou": part_iou([gt_box], pred_box, nocs),
            "iou": part_iou(gt_posed, pose_box(pred_pose, pred_box), nocs),
            "gt_bbox_iou": part_iou(gt_posed, pose_box(pred_pose, gt_box), nocs)}


# ---- tensor versions used by the training losses (reference bbox_utils.py:64-72, 88-92) ------------------------------
def tensor_bbox_from_corners(corners, device=None):
    """(..., 2, 3) [min; max] -> (..., 8, 3) corner points (same order as bbox_from_corners), torch."""
    import torch
    c = corners if isinstance(corners, torch.Tensor) else torch.as_tensor(np.asarray(corners)).float()
    if device is not None:
        c = c.to(device)
    pts = [torch.stack([c[..., (i % 4) // 2, 0], c[..., i // 4, 1], c[..., i % 2, 2]], dim=-1) for i in range(8)]
    return torch.stack(pts, dim=-2)


def yaxis_from_corners(corners, device=None):
    """Symmetric objects are compared along their axis only: the two corners with x and z zeroed, (..., 2, 3)."""
    import torch
    c = corners if isinstance(corners, torch.Tensor) else torch.as_tensor(np.asarray(corners)).float()
    if device is not None:
        c = c.to(device)
    return c * torch.tensor((0.0, 1.0, 0.0), device=c.device).reshape((1,) * (c.dim() - 1) + (3,))
