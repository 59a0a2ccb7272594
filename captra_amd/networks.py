"""CoordinateNet, RotationNet and the per-frame pose update.

Mirrors the reference's network/models/networks.py: `CoordNet` (l.19-107),
`RotationRegressionBackbone` (l.110-141), `PartCanonNet` (l.144-240) — same constructor
arguments, sub-module names (state-dict keys) and input/output dict layouts.
MI355X specifics: canonicalisation is one kernel that emits both memory layouts the backbone
needs; the pose fit is one kernel with no host round trip; labels travel as int32.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .backbones import PointNet2Msg
from .blocks import RotationRegressor, get_point_mlp, run_point_mlp
from .fold import fold_conv_bn
from .pose_utils.part_dof_utils import convert_pred_rtvec_to_matrix, merge_reenact_canon_part_pose
from .pose_utils.pose_fit import part_fit_st_cn, part_fit_st_track
from .pose_utils.procrustes import (rot_around_yaxis_to_3d, scale_pts_mask, transform_pts_2d_mask,
                                    translate_pts_mask)


def _canonicalize(points, points_mean, pose, num_parts=1, want_planes=False):
    """R^T((points + mean) - t)/s for B*num_parts clouds -> (cn (Q,3,N), n3 (Q,N,3)[, planes (Q,3,pad256(N))])."""
    if points.is_cuda:
        return fused.canonicalize(points.float().contiguous(), points_mean.float(), pose["rotation"].float(),
                                  pose["translation"].float(), pose["scale"].float(), num_parts, want_planes=want_planes)
    raise RuntimeError("captra_amd networks run on the GPU only (no CPU path)")


class CoordNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        out_dim = cfg["network"]["backbone_out_dim"]
        self.backbone = PointNet2Msg(cfg, out_dim, net_type="camera", use_xyz_feat=True)
        self.num_parts = cfg["num_parts"]
        self.sym = cfg["obj_sym"]
        seg_dim = self.num_parts + cfg["obj"]["extra_dims"]
        self.seg_head = get_point_mlp(out_dim, seg_dim, [], acti="none", dropout=None)
        self.nocs_head = get_point_mlp(out_dim, 3 * self.num_parts, cfg["network"]["nocs_head_dims"],
                                       acti="sigmoid", dropout=None)
        self._cache = {}

    def _drop_cache(self):
        if self._cache:
            from .fold import bump_weights_version
            bump_weights_version()
        self._cache = {}

    def train(self, mode=True):
        if mode != self.training:       # eval() while already in eval mode keeps the folded weights (see _FoldCache.train)
            self._drop_cache()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._drop_cache()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._drop_cache()
        return super()._apply(fn, *a, **k)

    def _head_layers(self, device):
        """[seg conv, nocs hidden conv+BN, nocs out conv] folded, when the heads have the shape the one-launch tail covers
        (seg head = one conv; NOCS head = conv-BN-ReLU, conv, sigmoid); None otherwise."""
        key = "tail_layers"
        if key not in self._cache:
            seg, nocs = list(self.seg_head), [m for m in self.nocs_head if not isinstance(m, nn.Dropout)]
            ok = (len(seg) == 1 and isinstance(seg[0], nn.Conv1d) and len(nocs) == 5 and isinstance(nocs[0], nn.Conv1d)
                  and isinstance(nocs[1], nn.BatchNorm1d) and isinstance(nocs[2], nn.ReLU) and isinstance(nocs[3], nn.Conv1d)
                  and isinstance(nocs[4], nn.Sigmoid))
            self._cache[key] = ([fold_conv_bn(seg[0], None, device), fold_conv_bn(nocs[0], nocs[1], device),
                                 fold_conv_bn(nocs[3], None, device)] if ok else None)
        return self._cache[key]

    def _heads(self, feat):
        if (not self.training) and feat.is_cuda:
            return run_point_mlp(self.seg_head, feat, self._cache), run_point_mlp(self.nocs_head, feat, self._cache)
        return self.seg_head(feat), self.nocs_head(feat)

    def forward(self, input, test=False):
        """input: {'points' (B,3,N), 'points_mean' (B,3,1), 'canon_pose' {rotation (B,3,3),
        translation (B,3,1), scale (B,)}, [...]} -> {'seg' (B,P+e,N) softmax, 'nocs' (B,3P,N), 'points'}."""
        canon_pose = input["canon_pose"]
        # `_canon` / `_geom`: the canonicalised cloud and its geometry, when the caller already computed them
        cam_cn, cam_n3 = input["_canon"] if "_canon" in input else _canonicalize(input["points"], input["points_mean"], canon_pose)
        self.last_canon = (cam_cn, cam_n3)
        fused_tail = None
        if (not self.training) and cam_cn.is_cuda:
            head_layers = self._head_layers(cam_cn.device)
            if head_layers is not None:
                def fused_tail(x, layers):
                    all_layers = list(layers) + head_layers
                    if fused.coord_tail_supported(x, all_layers):
                        return fused.coord_tail(x, all_layers)            # (seg logits, sigmoid(nocs) - 0.5)
                    if fused.chain_bf16_supported(x, layers, head_layers):
                        return fused.mlp_chain_bf16_fused(x.contiguous(), layers, head_layers)      # one launch, nothing in between stored
                    if fused.mlp_dtype() == "bf16":
                        # bf16 mode: FP1 + conv1 leave the feature map as a bf16 point-major tensor both heads read
                        n = x.shape[2]
                        feat = fused.mlp_chain_bf16(x, layers, [fused.ACT_RELU] * len(layers), out_pm=True)
                        seg_logits = fused.pointwise_mlp_bf16pm(feat, head_layers[0], n, in_pm=True, out_pm=False)
                        hid = fused.pointwise_mlp_bf16pm(feat, head_layers[1], n, in_pm=True, out_pm=True, act=fused.ACT_RELU)
                        return seg_logits, fused.pointwise_mlp_bf16pm(hid, head_layers[2], n, in_pm=True, out_pm=False,
                                                                      act=fused.ACT_SIGMOID_M05)
                    seg_logits, nocs = self._heads(fused.mlp_chain3(x, layers, fused.ACT_RELU))
                    return seg_logits, nocs - 0.5
        out = self.backbone(cam_cn, input_n3=cam_n3, geom=input.get("_geom"), finish=fused_tail)
        if fused_tail is not None:
            seg_logits, nocs_m05 = out
        else:
            seg_logits, nocs = self._heads(out)
            nocs_m05 = nocs - 0.5
        # The fused read-out's int32 labels travel WITH the prediction under the private key "_labels_i32" (EvalTrackModel pops it
        # before the dict leaves the step: tied to this forward's seg / nocs whatever ran in between; `last_labels_i32` mirrors it
        # for callers of the module alone); the dict's public keys stay the reference's (networks.py:44-52).  The label is the first index of the largest
        # LOGIT; the reference's argmax of the softmax output differs only when distinct logits round to equal probabilities
        # (lower index there) or a logit is NaN (torch.argmax returns the NaN's index).
        self.last_labels_i32 = None
        if ((not self.training) and seg_logits.is_cuda and seg_logits.dim() == 3 and seg_logits.shape[1] <= 8
                and seg_logits.dtype == torch.float32 and not (torch.is_grad_enabled() and seg_logits.requires_grad)):
            seg, self.last_labels_i32 = fused.seg_softmax_argmax(seg_logits.contiguous())      # softmax + arg max + int32 labels: one launch
            pred = {"seg": seg, "nocs": nocs_m05, "points": cam_cn, "_labels_i32": self.last_labels_i32}
        else:
            pred = {"seg": F.softmax(seg_logits, dim=1), "nocs": nocs_m05, "points": cam_cn}
        if "gt_part" in input:
            pred["part"] = self._fit_with_gt_rotation(input, pred, test)
        return pred

    def _fit_with_gt_rotation(self, input, pred, test):
        """Scale/translation from predicted NOCS under the ground-truth rotation
        (training-time diagnostics branch, reference networks.py:54-105)."""
        P = self.num_parts
        labels = torch.argmax(pred["seg"], dim=-2) if test else input["labels"]
        rotation = input["gt_part"]["rotation"]
        npcs = pred["nocs"].reshape(len(pred["nocs"]), P, 3, -1)
        cam_points = (input["points"] + input["points_mean"]).unsqueeze(1).repeat(1, P, 1, 1)
        eye = torch.cat([torch.eye(P), torch.zeros(2, P)], dim=0).to(npcs.device)
        mask = eye[labels].transpose(-1, -2)
        valid = (mask.sum(dim=-1) > 0).float()
        init_part = input["init_part"]
        if self.sym:
            canon_cam = torch.matmul(rotation.transpose(-1, -2), cam_points)
            rot_2d, _ = transform_pts_2d_mask(npcs[..., [0, 2], :].transpose(-1, -2),
                                              canon_cam[..., [0, 2], :].transpose(-1, -2), mask.unsqueeze(-1))
            rotated = torch.matmul(rotation, torch.matmul(rot_around_yaxis_to_3d(rot_2d), npcs))
        else:
            rotated = torch.matmul(rotation, npcs)
        smask = mask.unsqueeze(-2)

        def center(x):
            c = torch.sum(x * smask, dim=-1, keepdim=True) / torch.clamp(torch.sum(smask, dim=-1, keepdim=True), min=1.0)
            return (x - c.detach()) * smask

        def keep_finite(new, old):
            bad = ~torch.isfinite(new)
            return torch.where(bad, old, new)

        scale = scale_pts_mask(center(rotated), center(cam_points), smask)
        scale = keep_finite(valid * scale + (1.0 - valid) * init_part["scale"], init_part["scale"])
        use_scale = scale if test else input["gt_part"]["scale"]
        trans = translate_pts_mask(use_scale[..., None, None] * rotated, cam_points, mask.unsqueeze(-1))
        v = valid[..., None, None]
        trans = v * trans + (1.0 - v) * init_part["translation"]
        bad = ~torch.isfinite(trans.sum((-1, -2)))[..., None, None]
        trans = torch.where(bad, init_part["translation"], trans)
        return {"rotation": rotation, "scale": scale, "translation": trans}


class RotationRegressionBackbone(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_parts = cfg["num_parts"]
        self.encoder = PointNet2Msg(cfg, cfg["network"]["backbone_out_dim"], use_xyz_feat=False)
        self.sym = cfg["obj_sym"]
        self.pose_pred = RotationRegressor(cfg["network"]["backbone_out_dim"], self.num_parts, symmetric=self.sym)
        self.cfg = cfg
        self._default = None

    def raw_point_rtvec(self, cam, cam_n3=None, geom=None):
        """cam (B*P,3,N) -> (B*P,R,N): backbone + rotation head p on the clouds of part p (fused read-out path:
        captra_rot_pool_compose does the per-point normalisation, the masked mean and the pose algebra)."""
        finish = None
        if fused.mlp_dtype() == "bf16" and cam.is_cuda and not self.training:
            # bf16 mode: FP1 + conv1 in one launch, the feature map goes to the heads as a bf16 point-major tensor
            def finish(x, layers):
                if fused.chain_bf16_supported(x, layers):
                    return fused.mlp_chain_bf16_fused(x.contiguous(), layers)
                return fused.mlp_chain3(x, layers, fused.ACT_RELU)
        return self.pose_pred.raw_diag(self.encoder(cam, input_n3=cam_n3, geom=geom, finish=finish))

    def forward(self, cam, cam_labels, cam_n3=None, geom=None):
        """cam (B,3,N), cam_labels (B,N) -> {'rtvec' (B,P,D) masked mean, 'point_rtvec' (B,P,D,N)}."""
        feat = self.encoder(cam, input_n3=cam_n3, geom=geom)
        P = self.num_parts
        labels = cam_labels.long()
        part_mask = (labels.unsqueeze(1) == torch.arange(P, device=labels.device).view(1, P, 1)).float().unsqueeze(-2)
        valid = (part_mask.sum(dim=(-1, -2)) > 0).float().unsqueeze(-1)                 # (B,P,1)
        raw = self.pose_pred(feat)                                                     # (B,P,D,N)
        pooled = (raw * part_mask).sum(-1) / torch.clamp_min(part_mask.sum(-1), 1.0)  # (B,P,D)
        if self._default is None or self._default.device != raw.device:   # built once: no host->device copy per frame
            d = torch.tensor((0.0, 1.0, 0.0)) if self.sym else torch.eye(3).reshape(-1)
            self._default = d.to(raw.device).reshape(1, 1, -1)
        pooled = valid * pooled + (1.0 - valid) * self._default
        return {"rtvec": pooled, "point_rtvec": raw}


class PartCanonNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.type = cfg["network"]["type"]
        self.regress_net = RotationRegressionBackbone(cfg)
        self.device = cfg["device"]
        self.num_parts = cfg["num_parts"]
        self.sym = cfg["obj_sym"]
        self.tree = cfg["obj_tree"]
        self.root = [i for i in range(self.num_parts) if self.tree[i] == -1][0]
        self.cfg = cfg
        self.return_point_rotation = False

    def forward(self, input, test_mode=False):
        """input: {'points' (B,3,N), 'points_mean' (B,3,1), 'state': {'part': pose}, 'pred_labels',
        'pred_nocs' (B,P,3,N), ...} -> {'part': {'rotation' (B,P,3,3), 'scale' (B,P),
        'translation' (B,P,3,1)}, 'point_rotation' (B,P,N,3,3)}."""
        part_pose = input["state"]["part"]
        P = self.num_parts
        if "canon_pose" in input:
            canon_pose = input["canon_pose"]
        else:
            canon_pose = {k: part_pose[k].reshape((-1,) + part_pose[k].shape[2:]) for k in ("rotation", "translation", "scale")}
        eval_rnpcs = self.type == "rot_coord_track"
        cam_seg = input["pred_labels"] if eval_rnpcs else input["labels"]
        B = len(input["points"])

        # every part sees the whole cloud, canonicalised with that part's previous pose
        # `shared` = (canonicalised cloud, backbone geometry) of CoordNet, valid when this net's clouds are
        # the same clouds (one part: the part's previous pose IS the CoordNet's canonical pose)
        shared = input.get("shared_geometry") if P == 1 else None
        fast = (eval_rnpcs and test_mode and not self.return_point_rotation and not self.training and input["points"].is_cuda
                and fused.USE_ROT_READOUT)
        if fast and input.get("_raw") is not None:
            cam_cn, cam_n3, geom = input["points"], None, None      # the heads already ran (side stream): nothing to canonicalise
        elif shared is not None:
            (cam_cn, cam_n3), geom = shared
        else:
            cam_cn, cam_n3 = _canonicalize(input["points"], input["points_mean"], canon_pose, num_parts=P)
            geom = None
        if fast:
            # tracking fast path: one launch for per-point normalisation + masked mean + frame + R_prev * dR
            raw = input.get("_raw")                                                       # computed ahead on a side stream?
            if raw is None:
                raw = self.regress_net.raw_point_rtvec(cam_cn, cam_n3=cam_n3, geom=geom)   # (B*P,R,N), head p on cloud (b,p)
            labels_i32 = input.get("pred_labels_i32")
            if labels_i32 is None:
                labels_i32 = input["pred_labels"].int().contiguous()
            rotation = fused.rot_pool_compose(raw, labels_i32, part_pose["rotation"].float().contiguous(), self.sym)
            npcs = input["pred_nocs"].reshape(B, P, 3, -1).float().contiguous()
            # camera points = points + mean and "an invalid fit keeps the previous scale / translation" inside the launch
            scale, trans, _ = part_fit_st_track(labels_i32, npcs, input["points"].float().contiguous(), input["points_mean"], rotation,
                                                part_pose["scale"], part_pose["translation"], self.sym)
            return {"part": {"rotation": rotation, "scale": scale, "translation": trans}}
        seg_rep = cam_seg.unsqueeze(1).expand(-1, P, -1).reshape(B * P, -1)
        pred = self.regress_net(cam_cn, seg_rep, cam_n3=cam_n3, geom=geom)

        out = {"rotation": convert_pred_rtvec_to_matrix(pred["rtvec"], self.sym)}       # (B*P,P,3,3)
        if self.return_point_rotation or not test_mode or self.type == "rot":
            # per-point rotations are only consumed by the RotationNet experiment's losses (training and its evaluation
            # pass); skipped while tracking
            out["point_rotation"] = convert_pred_rtvec_to_matrix(pred["point_rtvec"].transpose(-1, -2), self.sym)
        diag = torch.arange(P, device=out["rotation"].device)
        for key in list(out.keys()):                                                     # head p on cloud p
            raw = out[key].reshape((B, P) + out[key].shape[1:])
            out[key] = raw[:, diag, diag]

        if self.type == "rot":
            final_pose = merge_reenact_canon_part_pose(part_pose, out)
            for key in ("translation", "scale"):
                final_pose[key] = input["gt_part"][key].detach().clone()
        elif eval_rnpcs:
            rotation = merge_reenact_canon_part_pose(part_pose, out)["rotation"]
            labels = input["pred_labels"] if test_mode else input["labels"]
            fit_rot = rotation if test_mode else input["gt_part"]["rotation"]
            npcs = input["pred_nocs"].reshape(B, P, 3, -1).float().contiguous()
            cam_points = (input["points"] + input["points_mean"]).float().contiguous()        # (B,3,N)
            scale, trans, valid = part_fit_st_cn(labels.int().contiguous(), npcs, cam_points,
                                                 fit_rot.float().contiguous(), self.sym)
            # invalid fits (<= 3 points, non-finite) keep the previous scale / translation
            # (reference networks.py:230-232 blends with a 0/1 float mask; `where` does the same on
            # finite values and does not let a NaN leak through 0 * NaN)
            final_pose = {"rotation": rotation,
                          "scale": torch.where(valid, scale, part_pose["scale"]),
                          "translation": torch.where(valid[..., None, None], trans, part_pose["translation"])}
        else:
            raise ValueError(f"unsupported network type {self.type}")

        ret = {"part": final_pose}
        if "point_rotation" in out:
            ret["point_rotation"] = out["point_rotation"]
        return ret
