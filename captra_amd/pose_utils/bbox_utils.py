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


def eval_instance_part_iou(gt_corners, pred_corners, gt_pose: dict, pred_pose: dict, nocs: bool = False, sym: bool = False) -> dict:
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


# ---- the reference's batch-level names (bbox_utils.py:75-85, 95-104, 107-125, 128-158, 160-198) --------------------------
def np_bbox_from_corners(corners):
    """(..., 2, 3) -> (..., 8, 3) torch tensor (the reference returns a CPU float tensor here despite the name)."""
    import torch
    return torch.from_numpy(bbox_from_corners(np.asarray(corners, dtype=np.float32)))


def get_posed_bbox_from_part(part_model: dict, corners):
    """Part poses (B,P,...) tensors + canonical corners (B,P,2,3) -> posed box points (B,P,8,3) numpy array."""
    from .part_dof_utils import pose_with_part
    return pose_with_part(part_model, tensor_bbox_from_corners(corners.detach(), corners.device)).detach().cpu().numpy()


def get_pred_nocs_corners(pred_seg, nocs_pred, num_parts: int) -> np.ndarray:
    """Predicted labels (B,N), the points' own-part NOCS (B,N,3) -> (B,P,2,3): per part the symmetric extent
    [-max|x|, +max|x|] of its points (zeros for an empty part)."""
    lab = pred_seg.detach().cpu().numpy() if hasattr(pred_seg, "detach") else np.asarray(pred_seg)
    nocs = nocs_pred.detach().cpu().numpy() if hasattr(nocs_pred, "detach") else np.asarray(nocs_pred)
    out = np.zeros((len(lab), num_parts, 2, 3), np.float64)
    for b in range(len(lab)):
        for p in range(num_parts):
            sel = lab[b] == p
            if sel.any():
                size = np.abs(nocs[b][sel]).max(axis=0)
                out[b, p] = np.stack([-size, size])
    return out


def calc_part_iou_list(gt_bbox_list, pred_bbox, separate="both", nocs=False):
    """Best IoU of every predicted part box against the candidate ground-truth boxes: boxes (B,P,8,3) ->
    ({part: batch mean}, {part: (B,) values}); separate=True / False picks one of the two like the reference."""
    to_np = lambda x: x if isinstance(x, np.ndarray) else x.detach().cpu().numpy()
    gts, pred = [to_np(g) for g in gt_bbox_list], to_np(pred_bbox)
    per = {p: np.array([part_iou([g[b] for g in gts], pred[b], nocs)[p] for b in range(pred.shape[0])]) for p in range(pred.shape[1])}
    mean = {p: np.mean(v) for p, v in per.items()}
    return per if separate is True else mean if separate is False else (mean, per)


def eval_single_part_iou(gt_corners, pred_corners, gt_pose: dict, pred_pose: dict, separate=False, nocs: bool = False, sym: bool = False):
    """Batch-level box IoUs, the reference's signature: corners (B,P,2,3) tensors, poses dicts of (B,P,...) tensors ->
    {'npcs_iou', 'iou', 'gt_bbox_iou'} of {part: value}: batch means (separate=True), per-instance arrays (False) or both."""
    import torch
    gc = gt_corners.detach().cpu().numpy()
    pc = pred_corners.detach().cpu().numpy() if isinstance(pred_corners, torch.Tensor) else np.asarray(pred_corners)
    g = {k: v.detach().cpu().numpy() for k, v in gt_pose.items()}
    q = {k: v.detach().cpu().numpy() for k, v in pred_pose.items()}
    rows = [eval_instance_part_iou(gc[b], pc[b], {k: v[b] for k, v in g.items()}, {k: v[b] for k, v in q.items()}, nocs=nocs, sym=sym)
            for b in range(len(gc))]
    per = {name: {p: np.array([r[name][p] for r in rows]) for p in range(gc.shape[1])} for name in ("npcs_iou", "iou", "gt_bbox_iou")}
    mean = {name: {p: np.mean(v) for p, v in d.items()} for name, d in per.items()}
    return mean if separate is True else per if separate is False else (mean, per)
