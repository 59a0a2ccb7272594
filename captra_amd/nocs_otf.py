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

def to_host(t: torch.Tensor) -> np.ndarray:
    """Device -> host through pinned memory and an event on the CURRENT stream.  A plain `.cpu()` lands in pageable memory,
    which the runtime serves with a staged copy that waits for the whole device — with two lanes of trajectories on two
    streams that wait is what kept one lane's sampler from running under the other lane's networks."""
    if not t.is_cuda:
        return t.numpy()
    buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    ev.synchronize()
    return buf.numpy()


def to_device(a, dev, dtype=None) -> torch.Tensor:
    """Host array -> device through pinned memory, asynchronously on the current stream (same reason as to_host)."""
    t = torch.as_tensor(np.ascontiguousarray(a)) if not isinstance(a, torch.Tensor) else a
    if dtype is not None:
        t = t.to(dtype)
    if torch.device(dev).type != "cuda":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


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


def proj_corners_batch(height: int, width: int, centers, radii, intrinsics=NOCS_REAL_INTRINSICS) -> np.ndarray:
    """proj_corners for B instances at once -> (B, 2, 2) int64.  Same float64 operations per instance; the 3x3 projection is
    written out (u = K00 x + K01 y + K02 z, ... with z = 1 after the homogeneous division), so nothing depends on how a BLAS
    orders a 3-term sum."""
    c = np.asarray(centers, np.float64).reshape(-1, 3)
    r = np.maximum(np.asarray(radii, np.float64).reshape(-1, 1), 0.05)
    lo, hi = c - r, c + r
    sel = np.array([[x, y, z] for y in (0, 1) for x in (0, 1) for z in (0, 1)])                 # the 8 corners: lo / hi per axis
    box = np.where(sel[None] == 0, lo[:, None, :], hi[:, None, :]) * 1000.0                       # (B, 8, 3)
    homog = -box / box[:, :, 2:3]
    homog[:, :, 2] = -homog[:, :, 2]
    K = np.asarray(intrinsics, np.float64)
    u = (K[0, 0] * homog[:, :, 0] + K[0, 1] * homog[:, :, 1]) + K[0, 2] * homog[:, :, 2]
    v = (K[1, 0] * homog[:, :, 0] + K[1, 1] * homog[:, :, 1]) + K[1, 2] * homog[:, :, 2]
    rows, cols = height - v.astype(np.int32), u.astype(np.int32)
    out = np.stack([np.stack([rows.min(1), cols.min(1)], -1), np.stack([rows.max(1), cols.max(1)], -1)], 1).astype(np.int64)
    out[:, 0] = np.maximum(out[:, 0], 0)
    out[:, 1] = np.minimum(out[:, 1], np.array([height - 1, width - 1]))
    return out


_KINV_CACHE: dict = {}


def _kinv(intrinsics) -> np.ndarray:
    """Inverse intrinsics as 9 doubles (cached per matrix: numpy.linalg.inv costs 30 us of the loop's critical host path)."""
    K = np.asarray(intrinsics, np.float64)
    key = K.tobytes()
    if key not in _KINV_CACHE:
        _KINV_CACHE[key] = np.linalg.inv(K).reshape(9)
    return _KINV_CACHE[key]


def _device_fps(points_f32: torch.Tensor, num: int) -> torch.Tensor:
    from . import fused
    res = fused.fps_gather(points_f32.reshape(1, -1, 3).contiguous(), num)
    if res is None:   # beyond the register-resident kernel: the drop-in op
        from .pointnet_lib import pointnet2_utils as pn
        return pn.furthest_point_sample(points_f32.reshape(1, -1, 3).contiguous(), num).reshape(-1).long()
    return res[0].reshape(-1).long()


def crop_candidates(depth: torch.Tensor, mask: torch.Tensor, center, radius: float, num_points: int,
                    intrinsics=NOCS_REAL_INTRINSICS, _depth_retry: int = 0):
    """Everything of the crop BEFORE the furthest-point sampling (nocs_data_process.py:92-109, 151-163;
    data_utils.py:146-152): -> (pts (n,3) float64 back-projected pixels, raw_mask (n,) bool, idx (c,) long = the ball
    members, list-doubled up to num_points, perm (5*num_points,) long or None = the thinning of an over-long list).
    The cloud handed to the sampler is pts[idx] (or pts[idx][perm]) as float32."""
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
        return crop_candidates(depth, mask, center, float(radius) * 1.2, num_points, intrinsics, _depth_retry + 1)
    while idx.numel() < num_points:
        idx = torch.cat([idx, idx])
    perm = None
    if idx.numel() > 5 * num_points:                                # data_utils.py:146-152
        perm = torch.from_numpy(np.random.permutation(idx.numel())[:5 * num_points]).to(dev)
    return pts, raw_mask, idx, perm


def _candidate_cloud(pts, idx, perm) -> torch.Tensor:
    cand = pts[idx]
    return (cand[perm] if perm is not None else cand).float()


def crop_ball_from_depth(depth: torch.Tensor, mask: torch.Tensor, center, radius: float, num_points: int,
                         intrinsics=NOCS_REAL_INTRINSICS, fps_fn=_device_fps):
    """depth (H,W) integer millimetres, mask (H,W) bool (the tracked instance), both on the compute device
    -> (points (num_points,3) float64 camera frame, obj_mask (num_points,) bool).  nocs_data_process.py:92-109, 151-163."""
    pts, raw_mask, idx, perm = crop_candidates(depth, mask, center, radius, num_points, intrinsics)
    picked = fps_fn(_candidate_cloud(pts, idx, perm), num_points)
    idx = idx[perm[picked] if perm is not None else picked]
    return pts[idx], raw_mask[idx]


def _full_data(pts, obj, gt_pose: dict) -> dict:
    dev = pts.device
    rot = torch.as_tensor(np.asarray(gt_pose["rotation"], np.float64).reshape(3, 3), device=dev)
    trans = torch.as_tensor(np.asarray(gt_pose["translation"], np.float64).reshape(1, 3), device=dev)
    scale = float(np.asarray(gt_pose["scale"], np.float64).reshape(-1)[0])
    nocs = torch.zeros_like(pts)
    nocs[obj] = ((pts[obj] - trans) / scale) @ rot
    return {"points": pts, "labels": 1 - obj.long(), "nocs": nocs}


def full_data_from_depth(depth, mask, center, radius, gt_pose: dict, num_points: int, intrinsics=NOCS_REAL_INTRINSICS,
                         fps_fn=_device_fps) -> dict:
    """-> {'points' (N,3), 'labels' (N,) (0 = object, 1 = background), 'nocs' (N,3)} float64 / int64 device tensors;
    gt_pose {'rotation' (3,3), 'translation' (3,1), 'scale' ()} of the instance (nocs_data_process.py:43-50, 227-236)."""
    pts, obj = crop_ball_from_depth(depth, mask, center, radius, num_points, intrinsics, fps_fn)
    return _full_data(pts, obj, gt_pose)


CROP_CAP = 65536          # ball members kept per instance by the crop kernel (more: that instance takes the torch path)


def _full_data_batch_torch(frames, num_points: int, intrinsics) -> list:
    """full_data_batch with the candidate extraction in torch ops, one instance after the other (the restatement the crop
    kernel is tested against); the sampling is already one ragged launch."""
    from . import fused
    cands = [crop_candidates(d, m, c, r, num_points, intrinsics) for d, m, c, r, _ in frames]
    clouds = [_candidate_cloud(p, i, pm) for p, _, i, pm in cands]
    counts = [int(c.shape[0]) for c in clouds]
    dev = clouds[0].device
    padded = torch.zeros(len(clouds), max(counts), 3, dtype=torch.float32, device=dev)
    for b, c in enumerate(clouds):
        padded[b, :counts[b]] = c
    res = fused.fps_gather(padded, num_points, n_per_cloud=torch.tensor(counts, dtype=torch.int32, device=dev))
    out = []
    for b, ((pts, raw_mask, idx, perm), frame) in enumerate(zip(cands, frames)):
        picked = res[0][b].long() if res is not None else _device_fps(clouds[b], num_points)
        sel = idx[perm[picked] if perm is not None else picked]
        out.append(_full_data(pts[sel], raw_mask[sel], frame[4]))
    return out


def stack_full_data(items: list) -> dict:
    """list of per-instance dicts -> {'points' (B,N,3), 'labels' (B,N), 'nocs' (B,N,3)}."""
    return {k: torch.stack([it[k] for it in items]) for k in ("points", "labels", "nocs")}


def full_data_batch(frames, num_points: int, intrinsics=NOCS_REAL_INTRINSICS, use_kernel: bool = True, stacked: bool = False):
    """The re-crop of ALL trajectories of a tracking step: one crop launch (captra_crop_ball, a workgroup per instance),
    one host round trip for the member counts, one furthest-point-sampling launch (captra_fps_gather_ragged).
    frames: list of (depth, mask, center, radius, gt_pose) -> list of full_data_from_depth's dicts, as if it had been
    called once per trajectory in list order (the thinning permutations are drawn in that order; nothing else draws).
    Instances on a rare path — fewer than 10 ball members (radius growth), more than CROP_CAP — take the torch path.
    stacked=True returns one dict of (B, …) tensors instead of the list (no per-instance slicing when nothing was rare)."""
    dev = frames[0][0].device
    if not use_kernel or dev.type != "cuda":
        res = _full_data_batch_torch(frames, num_points, intrinsics)
        return stack_full_data(res) if stacked else res
    gt = {"rotation": np.stack([np.asarray(f[4]["rotation"], np.float64).reshape(3, 3) for f in frames]),
          "translation": np.stack([np.asarray(f[4]["translation"], np.float64).reshape(3) for f in frames]),
          "scale": np.array([float(np.asarray(f[4]["scale"], np.float64).reshape(-1)[0]) for f in frames])}
    return full_data_batch_arrays(torch.stack([f[0] for f in frames]), torch.stack([f[1] for f in frames]),
                                  np.stack([np.asarray(f[2], np.float64).reshape(3) for f in frames]),
                                  np.array([float(f[3]) for f in frames], np.float64), gt, num_points, intrinsics, stacked)


_KMAT_DEV: dict = {}


def _intrinsics_on_device(intrinsics, dev) -> torch.Tensor:
    """[K (9 doubles), K^-1 (9 doubles)] on the device, uploaded once per (matrix, device)."""
    K = np.asarray(intrinsics, np.float64)
    key = (K.tobytes(), str(dev))
    if key not in _KMAT_DEV:
        _KMAT_DEV[key] = torch.from_numpy(np.concatenate([K.reshape(9), _kinv(intrinsics)])).to(dev)
    return _KMAT_DEV[key]


_ARANGE: dict = {}


def _arange(n: int, dev) -> torch.Tensor:
    key = (n, str(dev))
    if key not in _ARANGE:
        _ARANGE[key] = torch.arange(n, device=dev)
    return _ARANGE[key]


def full_data_batch_arrays(depth, mask, centers, radii_in, gt, num_points: int, intrinsics=NOCS_REAL_INTRINSICS, stacked: bool = True,
                           pose_dev=None, gt_dev=None, defer=None, mean=None):
    """full_data_batch on already-stacked inputs (the track loop's form: no per-trajectory Python on the frame's critical
    path): depth (B,H,W), mask (B,H,W) device tensors; centers (B,3), radii_in (B,) float64 host arrays (radius as handed to
    full_data_from_depth: clamped to 0.05 inside); gt = {'rotation' (B,3,3), 'translation' (B,3), 'scale' (B,)} float64 host arrays.
    `pose_dev` = (translation (B,3) fp32, scale (B,) fp32, radius_factor) ON THE DEVICE instead of centers / radii_in (pass None
    for both): the crop's box, centre and radius are then derived there (captra_crop_box: the same float64 operations) and the
    pose never visits the host -- the stage's ONE round trip is the member counts.  `gt_dev`: the same three arrays as float64 DEVICE
    tensors (uploaded once with the trajectory), used instead of three uploads per frame.
    `defer` (with pose_dev; an int = upper bound of the candidate lists' length): NO round trip at all -- see the branch below; the
    result is then {'points_cn' (B,3,N) fp32 = points - `mean` (B,3), 'labels' (B,N), 'nocs_cn' (B,3,N) fp32, '_info'}: the
    frame's tensors in the networks' layouts (captra_otf_finish) and the device int32 word [a rare-path instance was met, longest
    list, ..] for the caller to read a frame late."""
    from . import _lib as L, fused
    dev = depth.device
    B = depth.shape[0]
    depth = depth.to(torch.int32).contiguous()
    mask = mask.to(torch.uint8).contiguous()
    _, H, W = depth.shape
    pts = torch.empty(B, CROP_CAP, 3, dtype=torch.float64, device=dev)
    obj = torch.empty(B, CROP_CAP, dtype=torch.uint8, device=dev)
    pix = torch.empty(B, CROP_CAP, dtype=torch.int32, device=dev)
    counts = torch.empty(B, 2, dtype=torch.int32, device=dev)
    if pose_dev is not None:
        trans_d, scale_d, factor = pose_dev
        trans_d = trans_d.reshape(B, 3).float().contiguous()
        scale_d = scale_d.reshape(B).float().contiguous()
        kk = _intrinsics_on_device(intrinsics, dev)
        box_d = torch.empty(B, 4, dtype=torch.int32, device=dev)
        ctr_d = torch.empty(B, 3, dtype=torch.float64, device=dev)
        rad_d = torch.empty(B, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            L.call("captra_crop_box", B, H, W, float(factor), L.ptr(trans_d), L.ptr(scale_d), kk.data_ptr(), L.ptr(box_d), L.ptr(ctr_d), L.ptr(rad_d))
            L.call("captra_crop_ball", B, H, W, CROP_CAP, L.ptr(depth), L.ptr(mask), L.ptr(box_d), L.ptr(ctr_d), L.ptr(rad_d),
                   kk.data_ptr() + 72, L.ptr(pts), L.ptr(obj), L.ptr(pix), L.ptr(counts))
    else:
        centers = np.ascontiguousarray(np.asarray(centers, np.float64).reshape(B, 3))
        radii_in = np.asarray(radii_in, np.float64).reshape(B)
        radii = np.maximum(radii_in, 0.05)
        boxes = proj_corners_batch(H, W, centers, radii_in, intrinsics).reshape(B, 4).astype(np.int32)
        kinv = _kinv(intrinsics)
        # ONE host-to-device copy for everything the crop kernel reads from the host: the doubles (centres, radii, K^-1), then
        # the int32 boxes (this stretch of host work sits between the pose round trip and the crop launch: the GPU waits for it)
        ndbl = 4 * B + 9
        blob = np.empty(8 * ndbl + 16 * B, np.uint8)
        blob[:8 * ndbl].view(np.float64)[:] = np.concatenate([centers.reshape(-1), radii, kinv])
        blob[8 * ndbl:].view(np.int32)[:] = boxes.reshape(-1)
        host = to_device(blob, dev)
        hp = host.data_ptr()
        with torch.cuda.device(dev):
            L.call("captra_crop_ball", B, H, W, CROP_CAP, L.ptr(depth), L.ptr(mask), hp + 8 * ndbl, hp,
                   hp + 8 * 3 * B, hp + 8 * 4 * B, L.ptr(pts), L.ptr(obj), L.ptr(pix), L.ptr(counts))
    if defer is not None and pose_dev is not None:
        # ---- NO round trip: the candidate lists' lengths, the member table and the ragged sampler's
        # per-cloud counts are derived from the crop's counts ON THE DEVICE; the sampler's padded stride is an upper bound the
        # caller knows without them (`defer` = that bound: the previous frame's longest list with a margin, at most 5 N).  What
        # the host would have decided from the counts -- an instance on a rare path (< 10 members: radius growth; a list longer
        # than the bound or than 5 N: thinning, which draws from numpy's generator on the host) -- comes back as a device word
        # [any rare instance, longest list] the caller reads one frame LATE and, when set, answers by running the frame again on
        # the synchronous path below.  Rare rows are clamped so that every launch stays inside its buffers; their output is unused.
        stride = int(min(max(int(defer), num_points), 5 * num_points))
        cand = torch.empty(B, stride, 3, dtype=torch.float32, device=dev)
        lens = torch.empty(B, dtype=torch.int32, device=dev)
        info = torch.empty(4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.call("captra_otf_candidates", B, CROP_CAP, stride, num_points, L.ptr(pts), L.ptr(counts), L.ptr(cand), L.ptr(lens), L.ptr(info))
        res = fused.fps_gather(cand, num_points, n_per_cloud=lens)
        if gt_dev is not None:
            rot, trans, scale = (gt_dev["rotation"].reshape(B, 3, 3).contiguous(), gt_dev["translation"].reshape(B, 3).contiguous(),
                                 gt_dev["scale"].reshape(B).contiguous())
        else:
            rot = to_device(np.ascontiguousarray(np.asarray(gt["rotation"], np.float64).reshape(B, 3, 3)), dev)
            trans = to_device(np.ascontiguousarray(np.asarray(gt["translation"], np.float64).reshape(B, 3)), dev)
            scale = to_device(np.ascontiguousarray(np.asarray(gt["scale"], np.float64).reshape(B)), dev)
        mean_d = (torch.zeros(B, 3, dtype=torch.float32, device=dev) if mean is None else mean.reshape(B, 3).float().contiguous())
        points_cn = torch.empty(B, 3, num_points, dtype=torch.float32, device=dev)
        labels = torch.empty(B, num_points, dtype=torch.int64, device=dev)
        nocs_cn = torch.empty(B, 3, num_points, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("captra_otf_finish", B, CROP_CAP, stride, num_points, L.ptr(pts), L.ptr(obj), L.ptr(counts), L.ptr(res[0]), L.ptr(mean_d),
                   L.ptr(rot), L.ptr(trans), L.ptr(scale), L.ptr(points_cn), L.ptr(labels), L.ptr(nocs_cn))
        return {"points_cn": points_cn, "labels": labels, "nocs_cn": nocs_cn, "_info": info}
    n_members = to_host(counts[:, 0].contiguous())                                               # the one sync of the stage
    if pose_dev is not None and any(int(c) < 10 or int(c) > CROP_CAP for c in n_members):
        # a rare-path instance (radius growth / more members than the table holds) takes the torch path, which wants the centre on the host
        cs = to_host(torch.cat([ctr_d, (float(factor) * scale_d.double()).reshape(B, 1)], dim=1))
        centers, radii_in = np.ascontiguousarray(cs[:, :3]), np.ascontiguousarray(cs[:, 3])

    def gt_of(b):
        return {"rotation": gt["rotation"][b], "translation": np.asarray(gt["translation"][b]).reshape(3, 1), "scale": gt["scale"][b]}

    # host: the candidate list of every instance as indices into its member table (list doubling = index modulo count,
    # thinning = a prefix of numpy's permutation), in trajectory order because the permutations consume numpy's generator
    lengths, perms, slow = [0] * B, {}, {}
    for b in range(B):
        c = int(n_members[b])
        if c < 10 or c > CROP_CAP:
            slow[b] = crop_candidates(depth[b], mask[b].bool(), centers[b], float(radii_in[b]), num_points, intrinsics)
            continue
        length = c
        while length < num_points:
            length *= 2
        if length > 5 * num_points:
            perms[b] = np.random.permutation(length)[:5 * num_points]
            length = 5 * num_points
        lengths[b] = length
    fast = [b for b in range(B) if lengths[b] > 0]
    out = [None] * B
    if fast:
        lens = [lengths[b] for b in fast]
        cnt_f = counts[fast, 0].long() if len(fast) < B else counts[:, 0].long()
        # candidate j of an instance = member (j mod count): computed on the device (no table upload); only a thinned
        # list needs its permutation prefix from the host
        table_d = torch.arange(max(lens), device=dev).unsqueeze(0) % cnt_f.unsqueeze(1)
        for i, b in enumerate(fast):
            if b in perms:
                table_d[i, :lens[i]] = to_device(perms[b], dev) % cnt_f[i]
        if len(fast) == B:
            pts_f, obj_f = pts, obj
        else:
            fidx = to_device(np.asarray(fast, np.int64), dev)
            pts_f, obj_f = pts[fidx], obj[fidx]
        cand = torch.gather(pts_f, 1, table_d.unsqueeze(-1).expand(-1, -1, 3)).float()
        res = fused.fps_gather(cand, num_points, n_per_cloud=to_device(np.asarray(lens, np.int32), dev))
        sel = torch.gather(table_d, 1, res[0].long())                                             # (F, N) member numbers
        P = torch.gather(pts_f, 1, sel.unsqueeze(-1).expand(-1, -1, 3))                            # (F, N, 3) float64
        O = torch.gather(obj_f, 1, sel).bool()
        sub = slice(None) if len(fast) == B else np.asarray(fast)
        if gt_dev is not None:
            pick = (lambda t: t) if len(fast) == B else (lambda t: t[fidx])
            rot, trans, scale = pick(gt_dev["rotation"].reshape(B, 3, 3)), pick(gt_dev["translation"].reshape(B, 1, 3)), pick(gt_dev["scale"].reshape(B))
        else:
            rot = to_device(np.ascontiguousarray(np.asarray(gt["rotation"], np.float64).reshape(B, 3, 3)[sub]), dev)
            trans = to_device(np.ascontiguousarray(np.asarray(gt["translation"], np.float64).reshape(B, 1, 3)[sub]), dev)
            scale = to_device(np.ascontiguousarray(np.asarray(gt["scale"], np.float64).reshape(B)[sub]), dev)
        nocs = torch.where(O.unsqueeze(-1), torch.bmm((P - trans) / scale.reshape(-1, 1, 1), rot), torch.zeros_like(P))
        labels = 1 - O.long()
        if stacked and len(fast) == B:
            return {"points": P, "labels": labels, "nocs": nocs}
        for i, b in enumerate(fast):
            out[b] = {"points": P[i], "labels": labels[i], "nocs": nocs[i]}
    for b, (p_all, raw_mask, idx, perm) in slow.items():
        picked = _device_fps(_candidate_cloud(p_all, idx, perm), num_points)
        sel_b = idx[perm[picked] if perm is not None else picked]
        out[b] = _full_data(p_all[sel_b], raw_mask[sel_b], gt_of(b))
    return stack_full_data(out) if stacked else out
