"""Training losses of CoordinateNet and RotationNet.

Restates the reference's network/models/loss.py with the same names, argument meaning and value conventions
(`compute_miou_loss` l.122-134, `compute_nocs_loss` l.42-81, `compute_sym_nocs_loss` l.84-119, `rot_trace_loss` l.151-177,
`rot_yaxis_loss` l.180-188, `trans_loss` l.191-198, `scale_loss` l.201-207, `compute_point_pose_loss` l.210-221,
`compute_part_dof_loss` l.224-236).  Plain differentiable torch: these run once per training step on (B, N)-sized
tensors and are not on the tracking path.  Random draws (the pair sample of the pair-wise-match loss) come from the CPU
generator in the reference's order, so seeded runs agree across devices.
"""
from __future__ import annotations

import torch

from .pose_utils.part_dof_utils import pose_with_part

EPS = 1e-6


def vector_loss(x: torch.Tensor, loss: str = "l2") -> torch.Tensor:
    """(..., D) -> (...): the vector's 2-norm ('l2' — not its square) or 1-norm ('l1')."""
    if loss not in ("l1", "l2"):
        raise ValueError(f"unsupported loss type {loss}")
    return torch.norm(x, p=2 if loss == "l2" else 1, dim=-1)


def choose_coord_by_label(x: torch.Tensor, labels, last_dim: int = 3) -> torch.Tensor:
    """x (B,N,last_dim*P), labels (B,N) in [0, P+1] -> (B,N,last_dim): the block of the point's own part; labels P and
    P + 1 (background classes) select zeros."""
    if labels is None:
        return x
    parts = x.shape[-1] // last_dim
    per_part = x.reshape(x.shape[:-1] + (parts, last_dim))
    padded = torch.cat([per_part, torch.zeros_like(per_part[..., :2, :])], dim=-2)          # (B,N,P+2,D)
    index = labels.long().reshape(labels.shape + (1, 1)).expand(labels.shape + (1, last_dim))
    return torch.gather(padded, -2, index).squeeze(-2)


def compute_sym_nocs_loss(nocs_pred, nocs_gt, labels, pwm_num: int = 128):
    """Symmetric objects (rotation about y is unobservable): a per-point distance on (y, x^2 + z^2) and a pair-wise-match
    term comparing the mutual distances of `pwm_num` sampled object points.  (B,N,3) x2, labels (B,N) -> two scalars."""
    yg, yp = nocs_gt[..., 1], nocs_pred[..., 1]
    rg = nocs_gt[..., 0] ** 2 + nocs_gt[..., 2] ** 2
    rp = nocs_pred[..., 0] ** 2 + nocs_pred[..., 2] ** 2
    dist = torch.sqrt((yg - yp) ** 2 + torch.abs(rg - rp) + 1e-8)
    mask = labels == 0
    has_obj = (mask.sum(dim=-1) > 0).float()
    dist_loss = torch.sum(dist * mask) / torch.clamp(mask.sum(), min=1.0)

    picks = []
    for b in range(len(labels)):                                   # one draw per cloud, in batch order, CPU generator
        cand = torch.where(labels[b] == 0)[0]
        if len(cand) == 0:
            cand = torch.where(labels[b] == 1)[0]
        picks.append(cand[torch.randint(len(cand), (pwm_num,)).to(cand.device)])
    picks = torch.stack(picks).unsqueeze(-1).expand(-1, -1, 3)      # (B,M,3)
    sg, sp = torch.gather(nocs_gt, 1, picks), torch.gather(nocs_pred, 1, picks)

    def pair_dist(p):
        return torch.norm(p.unsqueeze(-2) - p.unsqueeze(-3), p=2, dim=-1)

    pwm = torch.abs(pair_dist(sg) - pair_dist(sp)).mean(dim=(-1, -2))
    pwm = torch.sum(pwm * has_obj) / torch.clamp(has_obj.sum(), min=1.0)
    return dist_loss, pwm


def compute_nocs_loss(nocs_per_part, nocs_gt, labels=None, confidence=None, loss="l2", self_supervise=True,
                      per_instance=False, sym=False, pwm_num=128):
    """nocs_per_part (B,3P,N) or (B,3,N), nocs_gt (B,3,N), labels (B,N) -> scalar (or the symmetric pair of scalars)."""
    pred = nocs_per_part.transpose(-1, -2)
    gt = nocs_gt.transpose(-1, -2)
    conf = torch.ones(gt.shape[:-1], device=gt.device) if (confidence is None or not self_supervise) else confidence
    mask = None
    if labels is not None and pred.shape[-1] > 3:
        parts = pred.shape[-1] // 3
        pred = choose_coord_by_label(pred, labels, last_dim=3)
        mask = labels < parts
    if sym:
        return compute_sym_nocs_loss(pred, gt, labels, pwm_num=pwm_num)
    raw = vector_loss(pred - gt, loss=loss) * conf
    ret = torch.mean(raw) if mask is None else torch.sum(raw * mask) / max(torch.sum(mask), 1.0)
    ret = ret - 0.1 * torch.mean(torch.log(conf))
    return (ret, raw) if per_instance else ret


def compute_miou_loss(pred, labels, per_instance=False):
    """Soft IoU of the predicted class probabilities: pred (B,C,N), labels (B,N) -> 1 - mean IoU over clouds and classes."""
    prob = pred.transpose(-1, -2)
    onehot = torch.eye(prob.shape[-1], device=labels.device)[labels]
    inter = torch.sum(prob * onehot, dim=-2)
    union = torch.sum(prob + onehot, dim=-2) - inter
    miou = inter / (union + EPS)
    loss = 1.0 - torch.mean(miou)
    return (loss, miou) if per_instance else loss


def compute_hard_miou_loss(pred, gt, num_parts, per_instance=False):
    """IoU of two hard labelings (B,N) with classes 0..num_parts-1, as 1 - mean IoU (reference l.137-148)."""
    a = torch.eye(num_parts, device=gt.device)[gt]
    b = torch.eye(num_parts, device=pred.device)[pred]
    inter = torch.sum(a * b, dim=-2)
    miou = inter / (torch.sum(a + b, dim=-2) - inter + EPS)
    loss = 1.0 - torch.mean(miou)
    return (loss, miou) if per_instance else loss


def rot_trace_loss(rot1, rot2, metric="l1"):
    """'frob': squared Frobenius norm of R1 - R2; 'l1' / 'l2': |trace(R1 R2^T) - 3| or its square; 'exp_l1' / 'exp_l2':
    difference of the rotation vectors."""
    if metric in ("exp_l1", "exp_l2"):
        from .pose_utils.rotations import matrix_to_rotvec
        diff = matrix_to_rotvec(rot1) - matrix_to_rotvec(rot2)
        return torch.abs(diff) if metric == "exp_l1" else diff ** 2
    if metric == "frob":
        d = rot1 - rot2
        prod = torch.matmul(d, d.transpose(-1, -2))
        return prod[..., 0, 0] + prod[..., 1, 1] + prod[..., 2, 2]
    if metric in ("l1", "l2"):
        prod = torch.matmul(rot1, rot2.transpose(-1, -2))
        tr = prod[..., 0, 0] + prod[..., 1, 1] + prod[..., 2, 2]
        return torch.abs(tr - 3) if metric == "l1" else (tr - 3.0) ** 2
    raise ValueError(f"unsupported metric {metric}")


def rot_yaxis_loss(rot1, rot2, metric="l2"):
    diff = rot1[..., 1] - rot2[..., 1]
    if metric == "l2":
        return (diff ** 2).sum(-1)
    if metric == "l1":
        return torch.norm(diff, p=2, dim=-1)
    raise ValueError(f"unsupported metric {metric}")


def trans_loss(trans1, trans2, metric="l1"):
    d = trans1 - trans2
    if metric == "l2":
        return torch.sum(d ** 2, dim=(-1, -2))
    if metric == "l1":
        return torch.norm(d.reshape(d.shape[:-1]), p=2, dim=-1)
    raise ValueError(f"unsupported metric {metric}")


def scale_loss(scale1, scale2, metric="l1"):
    if metric == "l2":
        return (scale1 - scale2) ** 2
    if metric == "l1":
        return torch.abs(scale1 - scale2)
    raise ValueError(f"unsupported metric {metric}")


def compute_point_pose_loss(gt_pose, pred_pose, pts, metric="l1"):
    """Distance between the same canonical points (box corners) posed by the two part poses: pts (B,P,K,3)."""
    diff = pose_with_part(gt_pose, pts) - pose_with_part(pred_pose, pts)
    if metric == "l2":
        dist = torch.sum(diff ** 2, dim=-1)
    elif metric == "l1":
        dist = torch.norm(diff, p=2, dim=-1)
    else:
        raise ValueError(f"unsupported metric {metric}")
    return dist.mean(), dist


def compute_part_dof_loss(gt, pred, pose_loss_type, collapse=True):
    out = {"sloss": scale_loss(gt["scale"], pred["scale"], metric=pose_loss_type["s"]),
           "tloss": trans_loss(gt["translation"], pred["translation"], metric=pose_loss_type["t"]),
           "rloss": rot_trace_loss(gt["rotation"], pred["rotation"], metric=pose_loss_type["r"])}
    return {k: v.mean() for k, v in out.items()} if collapse else out
