"""Seeded synthetic clouds shared by tests, golden generation and bench.py."""
from __future__ import annotations

import numpy as np


def s_nocs(i: int, n_obj: int = 3277, n_bg: int = 819):
    """S-nocs(seed): y-axis cylinder (r=0.18, h=0.90, caps, area-uniform, jitter 0.003) labelled 0
    plus background points in the ball |x|<0.6 with y<-0.45 labelled 1; random permutation;
    mean-subtracted.  Returns (points (N,3) f32, labels (N,) i64, mean (3,) f32)."""
    rng = np.random.default_rng(1000 + i)
    r, h = 0.18, 0.90
    a_side, a_cap = 2 * np.pi * r * h, np.pi * r * r
    p_side = a_side / (a_side + 2 * a_cap)
    u = rng.random(n_obj)
    which = np.where(u < p_side, 0, np.where(u < p_side + (1 - p_side) / 2, 1, 2))
    th = rng.random(n_obj) * 2 * np.pi
    rad = np.where(which == 0, r, r * np.sqrt(rng.random(n_obj)))
    y = np.where(which == 0, (rng.random(n_obj) - 0.5) * h, np.where(which == 1, h / 2, -h / 2))
    obj = np.stack([rad * np.cos(th), y, rad * np.sin(th)], -1) + rng.normal(0, 0.003, (n_obj, 3))
    bg = []
    while len(bg) < n_bg:
        c = (rng.random((4 * n_bg, 3)) * 2 - 1) * 0.6
        c = c[(np.linalg.norm(c, axis=1) < 0.6) & (c[:, 1] < -0.45)]
        bg.extend(c.tolist())
    bg = np.asarray(bg[:n_bg])
    pts = np.concatenate([obj, bg], 0).astype(np.float32)
    lab = np.concatenate([np.zeros(n_obj, np.int64), np.ones(n_bg, np.int64)])
    perm = rng.permutation(len(pts))
    pts, lab = pts[perm], lab[perm]
    mean = pts.mean(0, keepdims=True).astype(np.float32)
    return (pts - mean).astype(np.float32), lab, mean[0]


def s_nocs_dup(i: int, n_unique: int = 3000, n: int = 4096):
    """S-nocs-dup: n_unique points of S-nocs tiled up to n (mirrors nocs_data_process.py:105-106)."""
    pts, lab, mean = s_nocs(i)
    idx = np.arange(n_unique)
    while len(idx) < n:
        idx = np.concatenate([idx, idx])
    idx = idx[:n]
    return pts[idx], lab[idx], mean


def s_arti(i: int, parts: int = 4, per_part: int = 1024):
    """S-arti(seed): `parts` boxes of per_part surface-ish points each, labels 0..parts-1."""
    rng = np.random.default_rng(2000 + i)
    pts, lab = [], []
    for p in range(parts):
        size = np.array([0.5, 0.12, 0.4]) * (0.8 + 0.4 * rng.random(3))
        centre = np.array([0.0, -0.3 + 0.2 * p, 0.05 * p])
        q = (rng.random((per_part, 3)) - 0.5) * size
        face = rng.integers(0, 3, per_part)
        sign = rng.integers(0, 2, per_part) * 2 - 1
        q[np.arange(per_part), face] = sign * size[face] / 2
        pts.append(q + centre)
        lab.append(np.full(per_part, p, np.int64))
    pts = np.concatenate(pts).astype(np.float32)
    lab = np.concatenate(lab)
    perm = rng.permutation(len(pts))
    pts, lab = pts[perm], lab[perm]
    mean = pts.mean(0, keepdims=True).astype(np.float32)
    return (pts - mean).astype(np.float32), lab, mean[0]


def s_uni(i: int, n: int = 16384):
    """S-uni16k(seed): n points uniform in [-0.5,0.5]^3."""
    rng = np.random.default_rng(3000 + i)
    return (rng.random((n, 3), dtype=np.float32) - 0.5).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# synthetic trajectories in the track loop's data contract (SURVEY.md §8b "Loop API")
# ---------------------------------------------------------------------------------------------
def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def make_trajectory(kind: str, batch: int, frames: int, seed: int = 0):
    """List over frames of frame dicts (torch tensors on the CPU).

    kind 'nocs': S-nocs clouds (P=1, labels 0 = object, 1 = background);
    kind 'arti': S-arti clouds (P=4 boxes).
    Canonical (NOCS) coordinates are the cloud itself scaled into the unit-diagonal box; the
    ground-truth pose of frame t is a smooth rigid motion about 1 m in front of the camera:
    cam = s * R_t * nocs + t_t.  'points' are mean-subtracted camera points."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    P = 1 if kind == "nocs" else 4
    canon, labels = [], []
    for b in range(batch):
        pts, lab, _ = (s_nocs if kind == "nocs" else s_arti)(seed * 100 + b)
        extent = np.linalg.norm(pts.max(0) - pts.min(0))
        canon.append(pts / extent)
        labels.append(lab)
    canon = np.stack(canon).astype(np.float32)            # (B,N,3) NOCS coordinates
    labels = np.stack(labels)
    N = canon.shape[1]
    scale = (0.30 + 0.05 * rng.random(batch)).astype(np.float32)
    base_t = np.stack([rng.normal(0, 0.05, batch), rng.normal(0, 0.05, batch), 1.0 + 0.1 * rng.random(batch)], -1)
    rate = rng.normal(0, 0.03, (batch, 2))
    vel = rng.normal(0, 0.01, (batch, 3))
    data = []
    for t in range(frames):
        rot = np.stack([_rot_y(0.4 + rate[b, 0] * t) @ _rot_x(0.2 + rate[b, 1] * t) for b in range(batch)]).astype(np.float32)
        trans = (base_t + vel * t).astype(np.float32)
        cam = scale[:, None, None] * np.einsum("bij,bnj->bni", rot, canon) + trans[:, None, :]
        if P > 1:
            # articulated: part p slides along its local x by a part-specific offset
            for p in range(P):
                off = 0.02 * p * (1 + 0.2 * t)
                cam[labels == p] += (rot[:, :, 0] * off)[np.nonzero(labels == p)[0]]
        mean = cam.mean(1, keepdims=True)
        part_poses = []
        for p in range(P):
            tp = trans.copy()
            if P > 1:
                tp = tp + rot[:, :, 0] * (0.02 * p * (1 + 0.2 * t))
            part_poses.append({"rotation": torch.from_numpy(rot.copy()),
                               "translation": torch.from_numpy(tp.astype(np.float32)).unsqueeze(-1),
                               "scale": torch.from_numpy(scale.copy())})
        corners = np.zeros((batch, P, 2, 3), np.float32)
        for b in range(batch):
            for p in range(P):
                sel = canon[b][labels[b] == p]
                corners[b, p, 0], corners[b, p, 1] = sel.min(0), sel.max(0)
        data.append({
            "points": torch.from_numpy((cam - mean).transpose(0, 2, 1).astype(np.float32).copy()),
            "labels": torch.from_numpy(labels.copy()),
            "nocs": torch.from_numpy(canon.transpose(0, 2, 1).copy()),
            "meta": {"path": [f"synthetic/inst{seed * 100 + b}/track0/{t:04d}.npz" for b in range(batch)],
                     "nocs2camera": part_poses,
                     "points_mean": torch.from_numpy(mean.transpose(0, 2, 1).astype(np.float32).copy()),
                     "nocs_corners": torch.from_numpy(corners)},
        })
    return data


# G9p fixture (tests/golden/make_golden_track_physical.py): tag -> (obj_category, obj_config, kind, frames, batch,
# weight seed, torch seed); trajectories from make_trajectory(kind, batch, frames, seed=7)
PHYSICAL_SETUPS = {
    "bottle": ("1", "obj_info_nocs.yml", "nocs", 9, 2, 21, 4321),
    "camera": ("3", "obj_info_nocs.yml", "nocs", 7, 2, 22, 4322),
    "laptop": ("5", "obj_info_nocs.yml", "nocs", 7, 2, 23, 4323),
    "drawers": ("drawers", "obj_info_sapien.yml", "arti", 7, 2, 24, 4324),
}
