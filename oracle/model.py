"""CPU restatement of the network + track loop (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Functional, driven by a plain state dict (key -> tensor) with the reference's parameter names, so
it shares no code with captra_amd.  Geometry goes through the C restatement (oracle/ops.py).
Two arithmetic modes for the shared MLPs:
  mlp='torch'  conv1x1 -> BatchNorm(eval) -> ReLU with torch CPU kernels, layer by layer, exactly
               the call sequence of the reference (pointnet_utils.py:242-245) — used to check
               this restatement against the golden vectors and as the timed CPU baseline;
  mlp='exact'  BatchNorm folded (float64 -> fp32), acc = bias, fmaf chain over k ascending in C —
               the arithmetic contract of the HIP kernels, for bit-exact GPU parity at small sizes.

Reference call stack restated here: EvalTrackModel.forward model.py:386-478 ->
CoordNet.forward networks.py:34-52 -> PointNet2Msg.forward backbones.py:55-69 ->
PointNetSetAbstractionMsg.forward pointnet_utils.py:213-250 / PointNetSetAbstraction :319-343 /
PointNetFeaturePropagation :265-299 -> PartCanonNet.forward networks.py:156-240 ->
RotationRegressor blocks.py:183-193 -> part_fit_st_no_ransac pose_fit.py:38-53.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O


# ---------------------------------------------------------------------------------------------
# shared MLP layers
# ---------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def fold(sd, conv, bn=None):
    """(wt (cin,cout) f32, bias (cout) f32) with eval-mode BatchNorm folded in float64."""
    w = _np(sd[conv + ".weight"]).astype(np.float64)
    w = w.reshape(w.shape[0], -1)
    b = _np(sd[conv + ".bias"]).astype(np.float64)
    if bn is not None:
        g, beta = _np(sd[bn + ".weight"]).astype(np.float64), _np(sd[bn + ".bias"]).astype(np.float64)
        mu, var = _np(sd[bn + ".running_mean"]).astype(np.float64), _np(sd[bn + ".running_var"]).astype(np.float64)
        inv = g / np.sqrt(var + 1e-5)
        w = w * inv[:, None]
        b = (b - mu) * inv + beta
    return np.ascontiguousarray(w.T).astype(np.float32), b.astype(np.float32)


def conv_bn_act(sd, x, conv, bn, mlp, act="relu"):
    """x numpy (B,cin,...) -> (B,cout,...)."""
    if mlp == "exact":
        wt, b = fold(sd, conv, bn)
        return O.pointwise_mlp(x, wt, b, {"none": 0, "relu": 1}[act])
    xt = torch.from_numpy(np.ascontiguousarray(x))
    shape = xt.shape
    xt = xt.reshape(shape[0], shape[1], -1)
    w = torch.as_tensor(_np(sd[conv + ".weight"])).reshape(-1, shape[1], 1)
    y = F.conv1d(xt, w, torch.as_tensor(_np(sd[conv + ".bias"])))
    if bn is not None:
        y = F.batch_norm(y, torch.as_tensor(_np(sd[bn + ".running_mean"])), torch.as_tensor(_np(sd[bn + ".running_var"])),
                         torch.as_tensor(_np(sd[bn + ".weight"])), torch.as_tensor(_np(sd[bn + ".bias"])), False, 0.1, 1e-5)
    if act == "relu":
        y = F.relu(y)
    return y.reshape((shape[0], y.shape[1]) + tuple(shape[2:])).numpy()


# ---------------------------------------------------------------------------------------------
# PointNet++ modules
# ---------------------------------------------------------------------------------------------
def sa_msg(sd, prefix, scfg, xyz_cn, feat, mlp):
    """PointNetSetAbstractionMsg: xyz_cn (B,3,N), feat (B,D,N) or None -> (new_xyz_cn, (B,D',S))."""
    xyz_n3 = np.ascontiguousarray(xyz_cn.transpose(0, 2, 1))
    S = scfg["npoint"]
    fps = O.furthest_point_sample(xyz_n3, S)
    new_n3 = np.take_along_axis(xyz_n3, fps[..., None].astype(np.int64), 1)
    outs = []
    for i, (radius, K) in enumerate(zip(scfg["radius_list"], scfg["nsample_list"])):
        idx = O.ball_query(radius, K, xyz_n3, new_n3)
        x = O.sa_group(feat, xyz_cn, new_n3, idx)                       # [feat, xyz - centre]
        for j in range(len(scfg["mlp_list"][i])):
            x = conv_bn_act(sd, x, f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}", mlp)
        outs.append(O.max_over_k(x))
    return np.ascontiguousarray(new_n3.transpose(0, 2, 1)), np.concatenate(outs, axis=1)


def sa_all(sd, prefix, nlayers, xyz_cn, feat, mlp):
    """PointNetSetAbstraction(group_all): cat([xyz, feat]) -> MLP -> max over points -> (B,D',1)."""
    x = np.concatenate([xyz_cn, feat], axis=1)[:, :, None, :]           # (B,3+D,1,N): M=1, K=N
    for j in range(nlayers):
        x = conv_bn_act(sd, x, f"{prefix}.mlp_convs.{j}", f"{prefix}.mlp_bns.{j}", mlp)
    return O.max_over_k(x)                                                # (B,D',1)


def fp(sd, prefix, nlayers, xyz1_cn, xyz2_cn, points1, points2, mlp):
    """PointNetFeaturePropagation (CUDA three_nn semantics)."""
    N = xyz1_cn.shape[2]
    if xyz2_cn.shape[2] == 1:
        interp = np.repeat(points2, N, axis=2)
        x = np.concatenate([points1, interp], axis=1) if points1 is not None else interp
    else:
        x = O.fp_interpolate_concat(np.ascontiguousarray(xyz1_cn.transpose(0, 2, 1)),
                                    np.ascontiguousarray(xyz2_cn.transpose(0, 2, 1)), points1, points2)
    for j in range(nlayers):
        x = conv_bn_act(sd, x, f"{prefix}.mlp_convs.{j}", f"{prefix}.mlp_bns.{j}", mlp)
    return x


def backbone(sd, prefix, pcfg, cloud_cn, use_xyz_feat, mlp="torch", want_levels=False):
    """PointNet2Msg.forward: cloud_cn (B,3,N) -> (B,128,N)."""
    l0_xyz = cloud_cn
    l0_points = cloud_cn if use_xyz_feat else None
    l1_xyz, l1_points = sa_msg(sd, f"{prefix}.sa1", pcfg["sa1"], l0_xyz, l0_points, mlp)
    l2_xyz, l2_points = sa_msg(sd, f"{prefix}.sa2", pcfg["sa2"], l1_xyz, l1_points, mlp)
    l3_points = sa_all(sd, f"{prefix}.sa3", len(pcfg["sa3"]["mlp"]), l2_xyz, l2_points, mlp)
    l3_xyz = np.zeros((cloud_cn.shape[0], 3, 1), np.float32)
    l2_up = fp(sd, f"{prefix}.fp3", len(pcfg["fp3"]["mlp"]), l2_xyz, l3_xyz, l2_points, l3_points, mlp)
    l1_up = fp(sd, f"{prefix}.fp2", len(pcfg["fp2"]["mlp"]), l1_xyz, l2_xyz, l1_points, l2_up, mlp)
    skip0 = np.concatenate([l0_xyz, l0_points], axis=1) if l0_points is not None else l0_xyz
    l0_up = fp(sd, f"{prefix}.fp1", len(pcfg["fp1"]["mlp"]), l0_xyz, l1_xyz, skip0, l1_up, mlp)
    out = conv_bn_act(sd, l0_up, f"{prefix}.conv1", f"{prefix}.bn1", mlp)
    if want_levels:
        return out, {"sa1": l1_points, "sa2": l2_points, "sa3": l3_points}
    return out


# ---------------------------------------------------------------------------------------------
# heads, pose algebra
# ---------------------------------------------------------------------------------------------
def coord_heads(sd, prefix, feat, mlp, nocs_hidden=1):
    """seg_head (one conv) and nocs_head (conv-BN-ReLU x nocs_hidden, conv, sigmoid) - 0.5."""
    seg = conv_bn_act(sd, feat, f"{prefix}.seg_head.0", None, mlp, act="none")
    x = feat
    for h in range(nocs_hidden):
        x = conv_bn_act(sd, x, f"{prefix}.nocs_head.{3 * h}", f"{prefix}.nocs_head.{3 * h + 1}", mlp)
    x = conv_bn_act(sd, x, f"{prefix}.nocs_head.{3 * nocs_hidden}", None, mlp, act="none")
    nocs = torch.sigmoid(torch.from_numpy(x)).numpy() - np.float32(0.5)
    e = torch.softmax(torch.from_numpy(seg), dim=1).numpy()
    return e, nocs


def _normalize(v):
    """rotations.py:302-314 on (...,3)."""
    mag = np.sqrt((v * v).sum(-1, keepdims=True))
    ok = (mag > 1e-8).astype(v.dtype)
    unit = v / np.maximum(mag, np.float32(1e-8))
    backup = np.zeros_like(v)
    backup[..., 0] = 1.0
    return unit * ok + backup * (1 - ok)


def _cross(u, v):
    return np.stack([u[..., 1] * v[..., 2] - u[..., 2] * v[..., 1], u[..., 2] * v[..., 0] - u[..., 0] * v[..., 2],
                     u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]], -1)


def ortho6d_to_matrix(p6):
    """rotations.py:330-343: columns x, y, z."""
    x = _normalize(p6[..., 0:3])
    z = _normalize(_cross(x, p6[..., 3:6]))
    y = _cross(z, x)
    return np.stack([x, y, z], -1)


def gram_schmidt(m):
    """rotations.py:356-372 on columns."""
    def proj(u, a):
        return ((u * a).sum(-1, keepdims=True) / np.maximum((u * u).sum(-1, keepdims=True), np.float32(1e-8))) * u
    a1, a2, a3 = m[..., :, 0], m[..., :, 1], m[..., :, 2]
    u1 = a1
    u2 = a2 - proj(u1, a2)
    u3 = a3 - proj(u1, a3) - proj(u2, a3)
    return np.stack([_normalize(u1), _normalize(u2), _normalize(u3)], -1)


def yaxis_to_matrix(v):
    """rotations.py:375-387: columns x, y, z with z = e_x × y."""
    y = _normalize(v)
    ex = np.zeros_like(y)
    ex[..., 0] = 1.0
    z = _normalize(_cross(ex, y))
    x = _cross(y, z)
    return np.stack([x, y, z], -1)


def rot_pool_compose(raw, labels, prev_rotation, sym):
    """The rotation read-out of PartCanonNet (reference networks.py:127-138 + 200-203, blocks.py:183-192, rotations.py:302-387,
    part_dof_utils.py:124-141): raw (B,P,D,N) = head p's per-point output on cloud (b,p) (D = 3: a y axis; D = 6: ortho6d),
    labels (B,N) (values >= P: background), prev_rotation (B,P,3,3) -> (R_prev . dR, dR), both (B,P,3,3).
    Per point: normalise / ortho6d -> matrix; masked mean over the points labelled p (default (0,1,0) / identity when the part
    has none); re-orthogonalise (from a y axis / Gram-Schmidt); compose with the previous rotation."""
    B, P, D, N = raw.shape
    delta = np.zeros((B, P, 3, 3), np.float32)
    for p in range(P):
        per_point = raw[:, p].transpose(0, 2, 1)                          # (B,N,D)
        rep = _normalize(per_point) if sym else ortho6d_to_matrix(per_point).reshape(B, N, 9)   # blocks.py:183-192
        mask = (labels == p).astype(np.float32)[..., None]               # (B,N,1)
        cnt = mask.sum(1)
        pooled = (rep * mask).sum(1) / np.maximum(cnt, np.float32(1.0))  # networks.py:133
        default = np.array([0, 1, 0], np.float32) if sym else np.eye(3, dtype=np.float32).reshape(-1)
        pooled = np.where(cnt > 0, pooled, default[None]).astype(np.float32)
        delta[:, p] = yaxis_to_matrix(pooled) if sym else gram_schmidt(pooled.reshape(B, 3, 3))
    return np.matmul(prev_rotation, delta).astype(np.float32), delta      # part_dof_utils.py:127


def rot_head(sd, prefix, feat, sym):
    """MLPConv1d 128->512->512->256->D with GroupNorm(C/2) (blocks.py:148-165): (B,128,N) -> (B,D,N)."""
    x = torch.from_numpy(feat)
    for li in range(4):
        conv = f"{prefix}.model.{3 * li}"
        w = torch.as_tensor(_np(sd[conv + ".weight"]))
        x = F.conv1d(x, w, torch.as_tensor(_np(sd[conv + ".bias"])))
        if li < 3:
            gn = f"{prefix}.model.{3 * li + 1}"
            C = x.shape[1]
            x = F.relu(F.group_norm(x, C // 2, torch.as_tensor(_np(sd[gn + ".weight"])), torch.as_tensor(_np(sd[gn + ".bias"])), 1e-5))
    return x.numpy()


def rot_head_exact(sd, prefix, feat):
    """Same head with the convs on the exact fmaf chain (GroupNorm stays torch)."""
    x = feat
    for li in range(4):
        wt, b = fold(sd, f"{prefix}.model.{3 * li}", None)
        x = O.pointwise_mlp(x, wt, b, 0)
        if li < 3:
            gn = f"{prefix}.model.{3 * li + 1}"
            xt = torch.from_numpy(x)
            x = F.relu(F.group_norm(xt, xt.shape[1] // 2, torch.as_tensor(_np(sd[gn + ".weight"])),
                                    torch.as_tensor(_np(sd[gn + ".bias"])), 1e-5)).numpy()
    return x


# ---------------------------------------------------------------------------------------------
# one tracking step and the loop
# ---------------------------------------------------------------------------------------------
def track_step(sd, cfg, points, points_mean, last_pose, mlp="torch", gt_labels=None):
    """points (B,3,N), points_mean (B,3,1), last_pose {'rotation' (B,P,3,3), 'translation' (B,P,3,1),
    'scale' (B,P)} numpy -> (new pose, {'seg','nocs','labels'})."""
    P, sym = cfg["num_parts"], cfg["obj_sym"]
    pcfg = cfg["pointnet"]["camera"]
    root = [i for i in range(P) if cfg["obj_tree"][i] == -1][0]
    B, _, N = points.shape
    mean = points_mean.reshape(B, 3)
    # CoordNet (networks.py:34-52)
    cam_cn, _ = O.canonicalize(points, mean, last_pose["rotation"][:, root], last_pose["translation"][:, root, :, 0],
                               last_pose["scale"][:, root], P=1)
    feat = backbone(sd, "npcs_net.backbone", pcfg, cam_cn, True, mlp)
    seg, nocs = coord_heads(sd, "npcs_net", feat, mlp, nocs_hidden=len(cfg["network"]["nocs_head_dims"]))
    labels = seg.argmax(axis=1) if gt_labels is None else gt_labels
    # PartCanonNet (networks.py:156-240): every part sees the whole cloud in its own frame
    cam_p, _ = O.canonicalize(points, mean, last_pose["rotation"].reshape(B * P, 3, 3),
                              last_pose["translation"].reshape(B * P, 3), last_pose["scale"].reshape(B * P), P=P)
    feat_r = backbone(sd, "net.regress_net.encoder", pcfg, cam_p, False, mlp)
    raws = []
    for p in range(P):                                                    # only head p on cloud (b,p) is used
        fr = np.ascontiguousarray(feat_r.reshape(B, P, 128, N)[:, p])
        head = f"net.regress_net.pose_pred.rtvec_head.{p}"
        raws.append(rot_head_exact(sd, head, fr) if mlp == "exact" else rot_head(sd, head, fr, sym))     # (B,D,N)
    rotation, _ = rot_pool_compose(np.stack(raws, axis=1), labels, last_pose["rotation"], sym)
    cam_points = (points + points_mean).astype(np.float32)
    scale, trans, valid = O.part_fit_st(labels, nocs.reshape(B, P, 3, N), cam_points, rotation, sym)
    v = valid.astype(bool)
    pose = {"rotation": rotation,
            "scale": np.where(v, scale, last_pose["scale"]).astype(np.float32),
            "translation": np.where(v[..., None, None], trans[..., None], last_pose["translation"]).astype(np.float32)}
    return pose, {"seg": seg, "nocs": nocs, "labels": labels}


def track(sd, cfg, data, init_pose, mlp="torch"):
    """data: list of frame dicts (tests/clouds.make_trajectory); init_pose numpy dict -> [pose]*T."""
    poses = [init_pose]
    aux = [None]
    for i in range(1, len(data)):
        f = data[i]
        pose, a = track_step(sd, cfg, f["points"].numpy(), f["meta"]["points_mean"].numpy(), poses[-1], mlp)
        poses.append(pose)
        aux.append(a)
    return poses, aux
