"""Seeded synthetic inputs of the tracking path: clouds and trajectories, depth frames for the on-the-fly
re-crop, and weights drawn in parameter-NAME order from numpy's default_rng -- so that the reference (at golden-generation
time), the product harnesses (`python -m captra_amd.track --synthetic`, `captra_amd.train`, bench.py) and the tests build
identical tensors from a seed, without any dataset or checkpoint being committed (SURVEY.md section 7 step 1, section 8c
adjustment 5).  tests/clouds.py and tests/weights.py re-export this module."""
from __future__ import annotations

import numpy as np
import torch

# ---- clouds and trajectories ---------------------------------------------------------------------------------------------
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

# G9p, second file (tests/golden/g9p_track_more.npz, same generator with --set more): the three rigid categories the first file
# does not hold (bowl, can: symmetric; mug: not), and a FIVE-trajectory bottle batch -- more trajectories than
# fused.SPLIT_K_MAX_TRAJECTORIES, so the free-running comparison pins the large-batch kernels (wave-per-centre SA scales, 64x64-tile
# dense layers) to the reference end to end, not only the few-trajectory forms
PHYSICAL_SETUPS_MORE = {
    "bowl": ("2", "obj_info_nocs.yml", "nocs", 7, 2, 25, 4325),
    "can": ("4", "obj_info_nocs.yml", "nocs", 7, 2, 43, 4326),
    "mug": ("6", "obj_info_nocs.yml", "nocs", 7, 2, 27, 4327),
    "bottle5": ("1", "obj_info_nocs.yml", "nocs", 5, 5, 29, 4328),
}

# ---- depth frames (the on-the-fly re-crop's input) -----------------------------------------------------------------------
def make_frame(seed: int, height: int = 480, width: int = 640, dr: float = 0.0, dc: float = 0.0):
    """Synthetic depth frame (uint16 mm) with an object blob in front of a wavy background, its mask, a predicted centre
    and radius, and an instance pose.  (dr, dc): the blob displaced by that many pixel rows / columns -- a moving object over
    the frames of a trajectory; the defaults are golden G11's frames."""
    rng = np.random.default_rng(seed)
    r, c = np.mgrid[0:height, 0:width]
    depth = 1500.0 + 80.0 * np.sin(r / 37.0) + 60.0 * np.cos(c / 53.0)
    cr, cc = 200 + 40 * rng.random() + dr, 300 + 60 * rng.random() + dc
    rr = np.sqrt((r - cr) ** 2 + (c - cc) ** 2)
    blob = rr < 70
    depth[blob] = 900.0 + 0.004 * rr[blob] ** 2 + 20.0 * np.sin(c[blob] / 9.0)
    depth[rng.random(depth.shape) < 0.02] = 0.0                       # holes
    depth = depth.astype(np.uint16)
    mask = blob & (rng.random(depth.shape) < 0.97)
    K = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])
    z = 0.95
    center = np.array([(cc - K[0, 2]) / K[0, 0] * z, ((height - cr) - K[1, 2]) / K[1, 1] * z, -z]) + 0.01 * rng.standard_normal(3)
    th = 0.4
    pose = {"rotation": np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]),
            "translation": center.reshape(3, 1) + 0.005, "scale": np.float64(0.31)}
    return depth, mask, center, pose


# ---- weights -----------------------------------------------------------------------------------------------------------
def make_state_dict(shapes: dict, seed: int = 7) -> dict:
    """shapes: {state-dict key: tuple shape}.  Kaiming-like conv weights (activations keep their
    scale through ~20 layers), non-trivial BatchNorm statistics and affine parameters."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        if leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "weight" and len(shape) == 1:      # BatchNorm / GroupNorm gamma
            v = rng.uniform(0.7, 1.3, shape)
        elif leaf == "weight":                          # conv weight (cout, cin, 1[,1])
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        elif leaf == "bias":
            v = rng.normal(0.0, 0.05, shape)
        else:
            raise KeyError(f"unexpected state-dict leaf {name}")
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out


def shapes_of(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def make_physical_state_dict(shapes: dict, seed: int, num_parts: int, sym: bool, kind: str = "nocs", nocs_gain: float = 4.6,
                             rot_head_gain: float = 0.05, nocs_mix: float = 0.1) -> dict:
    """Seeded weights under which the track loop TRACKS the synthetic trajectories (positive, converging scales; a
    translation estimate that does not inherit the previous frame's error; small frame-to-frame rotations), so that a
    free-running trajectory does not amplify rounding noise and can be held to the 1e-4 contract on every frame (golden
    G9p), and the timed trajectories of bench.py stay meaningful over hundreds of frames.

    `make_state_dict(shapes, seed)` with plants (every other tensor keeps its random values and still feeds every output):
      * CoordinateNet carries nine numbers per point through identity rows (BatchNorm identity on those rows; all of them
        are >= 0 on |x| < 1, so the ReLUs pass them unchanged): (x + 1) of the canonicalised input coordinates, and the
        cloud's bounding box as max(x) + 1 and 1 - min(x) -- taken by the max-pools of SA1 (ball), SA2 (ball) and SA3
        (group_all) over the coordinate FEATURES this backbone groups (use_xyz_feat), broadcast back by FP3 and carried
        down by FP2 / FP1's interpolation (weights sum to 1).  The NOCS output conv combines them into the box-centred
        coordinate u = x - (max + min) / 2 with gain `nocs_gain`: sigmoid(g u) - 0.5 ~ u (slope g/4 > 1 at 0, < 1 far out,
        so the fitted scale has an attracting fixed point), plus `nocs_mix` x its random weights on all 128 channels.  A
        box-centred prediction does not depend on where the previous pose put the cloud: translation errors do not persist;
      * the segmentation head reads the box-centred y the same way on top of `nocs_mix` x its random weights: `kind` "nocs"
        (S-nocs clouds: object above its background) splits at u_y = -0.24, "arti" (S-arti: four boxes stacked along y)
        labels by the nearest box centre -- every part keeps hundreds of points and few points sit near a decision
        boundary (the generator asserts a margin);
      * RotationNet: the rotation heads' output conv is scaled by `rot_head_gain` and biased to the identity rotation
        ((0,1,0) for symmetric objects, ortho6d (1,0,0,0,1,0) otherwise): dR = I + a small input-dependent rotation.
    """
    sd = make_state_dict(shapes, seed)

    def plant(conv, bn, rows):
        """rows: [(out channel, {in channel: weight}, bias)] -- the row is zeroed first, its BatchNorm made the identity."""
        w = sd[f"{conv}.weight"]
        for o, ins, bias in rows:
            w[o] = 0.0
            for i, v in ins.items():
                w[o, i] = v
            sd[f"{conv}.bias"][o] = bias
            if bn is not None:
                sd[f"{bn}.weight"][o] = 1.0
                sd[f"{bn}.bias"][o] = 0.0
                sd[f"{bn}.running_mean"][o] = 0.0
                sd[f"{bn}.running_var"][o] = 1.0

    def carry(conv, bn, n, src0=0):
        plant(conv, bn, [(k, {src0 + k: 1.0}, 0.0) for k in range(n)])

    B = "npcs_net.backbone"
    # SA1, scale 0 (input: coordinate features 0..2, relative xyz 3..5): x + 1 and 1 - x, then max over the ball
    plant(f"{B}.sa1.conv_blocks.0.0", f"{B}.sa1.bn_blocks.0.0",
          [(c, {c: 1.0}, 1.0) for c in range(3)] + [(3 + c, {c: -1.0}, 1.0) for c in range(3)])
    carry(f"{B}.sa1.conv_blocks.0.1", f"{B}.sa1.bn_blocks.0.1", 6)
    carry(f"{B}.sa1.conv_blocks.0.2", f"{B}.sa1.bn_blocks.0.2", 6)
    # SA2, scale 0 (input: the 320 SA1 features first, relative xyz last): carry, max over the ball
    for l in range(3):
        carry(f"{B}.sa2.conv_blocks.0.{l}", f"{B}.sa2.bn_blocks.0.{l}", 6)
    # SA3 group_all (input: xyz FIRST, then the 512 SA2 features): carry, max over all 128 points = the box
    carry(f"{B}.sa3.mlp_convs.0", f"{B}.sa3.mlp_bns.0", 6, src0=3)
    carry(f"{B}.sa3.mlp_convs.1", f"{B}.sa3.mlp_bns.1", 6)
    carry(f"{B}.sa3.mlp_convs.2", f"{B}.sa3.mlp_bns.2", 6)
    # FP3 (input: 512 SA2 features, then the 1024 broadcast SA3 features), FP2 (320 SA1 features, then FP3's 256)
    carry(f"{B}.fp3.mlp_convs.0", f"{B}.fp3.mlp_bns.0", 6, src0=512)
    carry(f"{B}.fp3.mlp_convs.1", f"{B}.fp3.mlp_bns.1", 6)
    carry(f"{B}.fp2.mlp_convs.0", f"{B}.fp2.mlp_bns.0", 6, src0=320)
    carry(f"{B}.fp2.mlp_convs.1", f"{B}.fp2.mlp_bns.1", 6)
    # FP1 (input: xyz, xyz, then FP2's 128): channels 0..2 = x + 1 of the point itself, 3..8 = the box
    plant(f"{B}.fp1.mlp_convs.0", f"{B}.fp1.mlp_bns.0",
          [(c, {c: 1.0}, 1.0) for c in range(3)] + [(3 + k, {6 + k: 1.0}, 0.0) for k in range(6)])
    carry(f"{B}.fp1.mlp_convs.1", f"{B}.fp1.mlp_bns.1", 9)
    carry(f"{B}.conv1", f"{B}.bn1", 9)
    carry("npcs_net.nocs_head.0", "npcs_net.nocs_head.1", 9)

    def centred(j, gain):
        """weights on channels (x_j + 1, max_j + 1, 1 - min_j) and the bias of gain * (x_j - (max_j + min_j) / 2)."""
        return {j: gain, 3 + j: -0.5 * gain, 6 + j: 0.5 * gain}, -gain

    w, b = sd["npcs_net.nocs_head.3.weight"], sd["npcs_net.nocs_head.3.bias"]
    w *= nocs_mix
    b *= nocs_mix
    for p in range(num_parts):
        for j in range(3):
            ins, bias = centred(j, nocs_gain)
            w[3 * p + j, :9] = 0.0
            for i, v in ins.items():
                w[3 * p + j, i] = v
            b[3 * p + j] = bias
    w, b = sd["npcs_net.seg_head.0.weight"], sd["npcs_net.seg_head.0.bias"]
    w *= nocs_mix
    b *= nocs_mix
    if kind == "nocs":                     # logit(part 0) - logit(background) = K (u_y - y0)
        K, y0 = 40.0, -0.24
        ins, bias = centred(1, K)
        for i, v in ins.items():
            w[0, i] += v
        b[0] += bias - K * y0
    else:                                  # nearest box centre along y: logit_p = K (c_p u_y - c_p^2 / 2)
        K = 200.0
        centres = [(-0.3 + 0.2 * p) / 1.2 for p in range(num_parts)]
        mid = sum(centres) / len(centres)
        for p, c in enumerate(centres):
            c -= mid
            ins, bias = centred(1, K * c)
            for i, v in ins.items():
                w[p, i] += v
            b[p] += bias - K * c * c / 2
    ident = (0.0, 1.0, 0.0) if sym else (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
    for p in range(num_parts):
        key = f"net.regress_net.pose_pred.rtvec_head.{p}.model.9"
        sd[f"{key}.weight"] *= rot_head_gain
        sd[f"{key}.bias"] = torch.tensor(ident, dtype=torch.float32)
    return sd


def make_otf_trajectory(batch: int, frames: int, seed: int = 0, step_px: float = 6.0):
    """A trajectory for the on-the-fly re-crop loop (`nocs_otf=True`, reference model.py:425-452): the frame dicts of
    make_trajectory('nocs', ...) carrying, per frame, the depth image / instance mask the loop re-crops
    (meta['pre_fetched']) and the blob's ground-truth pose (meta['nocs2camera']); trajectory b watches depth frame
    make_frame(seed + b) whose blob drifts by `step_px` pixels per time step.  Also the reference's path conventions
    (meta['path'] = .../<category>/<instance>/<track>/<frame>.npz, meta['ori_path'])."""
    data = make_trajectory("nocs", batch, frames, seed=seed)
    for t, f in enumerate(data):
        views = [make_frame(seed + b, dr=0.5 * step_px * t, dc=step_px * t) for b in range(batch)]
        f["meta"]["pre_fetched"] = {"depth": torch.from_numpy(np.stack([v[0].astype(np.int32) for v in views])),
                                    "mask": torch.from_numpy(np.stack([v[1] for v in views]))}
        f["meta"]["path"] = [f"synthetic/1/inst{seed + b}/track0/{t:04d}.npz" for b in range(batch)]
        f["meta"]["ori_path"] = [f"synthetic/scene_{seed + b}/{t:04d}_depth.png" for b in range(batch)]
        for p in f["meta"]["nocs2camera"]:
            p["rotation"] = torch.from_numpy(np.stack([v[3]["rotation"] for v in views])).float()
            p["translation"] = torch.from_numpy(np.stack([v[3]["translation"] for v in views])).float()
            p["scale"] = torch.tensor([float(v[3]["scale"]) for v in views])
    return data


# G15 fixture (tests/golden/make_golden_otf_loop.py): tag -> (frames, trajectory seed, weight seed, torch / numpy seed)
OTF_LOOP_SETUPS = {"a": (5, 1, 31, 5001), "b": (4, 2, 32, 5002)}
