"""The two training experiments of CAPTRA: CoordinateNet (`CanonCoordModel`) and RotationNet (`RotationModel`).

Mirrors the reference's network/models/model.py (`BaseModel` l.27-104, `CanonCoordModel` l.107-202, `RotationModel`
l.205-306): same constructor, `set_data / update / test / compute_loss`, `feed_dict / pred_dict / loss_dict` layouts and
loss-dict keys, so that `Trainer.update(data)` is a drop-in.  SURVEY.md §8f row 4: the forward pass runs the networks
in training mode — layer by layer under autograd, torch convolutions / BatchNorm with batch statistics over the HIP
sampling, ball-query, grouping and interpolation operators (whose backward kernels are captra_group_points_grad /
captra_three_interpolate_grad) — not the fused inference kernels.  Pinned by golden G12 (the reference's own update).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .loss import (compute_miou_loss, compute_nocs_loss, compute_part_dof_loss, compute_point_pose_loss, rot_trace_loss,
                   rot_yaxis_loss)
from .networks import CoordNet, PartCanonNet
from .pose_utils.bbox_utils import tensor_bbox_from_corners, yaxis_from_corners
from .pose_utils.part_dof_utils import (add_noise_to_part_dof, compute_parts_delta_pose, eval_part_full,
                                        part_model_batch_to_part)
from .utils import cvt_torch


class BaseModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_parts = int(cfg["num_parts"])
        self.num_joints = int(cfg["num_joints"])
        self.device = cfg["device"]
        self.loss_weights = cfg["loss_weight"]
        self.network_type = cfg["network"]["type"]
        raw = cfg["pose_perturb"]
        self.pose_perturb_cfg = {"type": raw["type"], "scale": raw["s"], "translation": raw["t"],
                                 "rotation": np.deg2rad(raw["r"])}
        self.sym = cfg["obj_sym"]
        self.pose_loss_type = cfg["pose_loss_type"]
        self.pwm_num = None if not self.sym else cfg["network"]["pwm_num"]
        self.cfg = cfg
        self.feed_dict, self.pred_dict, self.loss_dict = {}, {}, {}
        self.per_diff_dict = {}

    def record_per_diff(self, data, per_diff):
        """Per-instance error tables keyed '<instance>_<track>_<frame>' (reference model.py:42-49)."""
        from .utils import get_ith_from_batch
        for i, path in enumerate(data["meta"]["path"]):
            instance, track_num, frame_i = path.split(".")[-2].split("/")[-3:]
            self.per_diff_dict.setdefault(f"{instance}_{track_num}_{frame_i}", {}).update(get_ith_from_batch(per_diff, i))

    def prepare_poses(self, data):
        """Ground-truth part poses and their perturbed copy = the pose the networks canonicalise with (model.py:49-58)."""
        gt_part = part_model_batch_to_part(cvt_torch(data["meta"]["nocs2camera"], self.device), self.num_parts, self.device)
        init_part = add_noise_to_part_dof(gt_part, self.pose_perturb_cfg)
        if "crop_pose" in data["meta"]:
            crop = part_model_batch_to_part(cvt_torch(data["meta"]["crop_pose"], self.device), self.num_parts, self.device)
            for key in ("translation", "scale"):
                init_part[key] = crop[key]
        return gt_part, init_part

    def summarize_losses(self, loss_dict):
        total = 0
        for key, weight in self.loss_weights.items():
            if key in loss_dict:
                total = total + loss_dict[key] * weight
        loss_dict["total_loss"] = total
        self.loss_dict = loss_dict

    def _gt_box(self, meta):
        corners = meta["nocs_corners"].float().to(self.device)
        return yaxis_from_corners(corners) if self.sym else tensor_bbox_from_corners(corners)

    def _pose_terms(self, gt_part, pred_part, init_part, loss_dict, per_instance=False):
        diff, per = eval_part_full(gt_part, pred_part, per_instance=per_instance, yaxis_only=self.sym)
        init_diff, init_per = eval_part_full(gt_part, init_part, per_instance=per_instance, yaxis_only=self.sym)
        loss_dict.update(diff)
        loss_dict.update({f"init_{k}": v for k, v in init_diff.items()})
        loss_dict.update(compute_part_dof_loss(gt_part, pred_part, self.pose_loss_type))
        return per, init_per


class CanonCoordModel(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.net = CoordNet(cfg)
        self.tree = cfg["obj_tree"]
        self.root = [p for p in range(len(self.tree)) if self.tree[p] == -1][0]

    def set_data(self, data):
        self.feed_dict = {"meta": data["meta"], "labels": data["labels"].long().to(self.device),
                          "points": data["points"].float().to(self.device), "nocs": data["nocs"].float().to(self.device),
                          "points_mean": data["meta"]["points_mean"].float().to(self.device)}

    def prepare_data(self):
        gt_part, init_part = self.prepare_poses(self.feed_dict)
        self.feed_dict["canon_pose"] = {k: init_part[k][:, self.root] for k in ("rotation", "translation", "scale")}
        self.feed_dict["init_part"] = init_part
        self.feed_dict["gt_part"] = gt_part

    def compute_loss(self, test=False):
        feed, pred = self.feed_dict, self.pred_dict
        loss_dict = {"seg_loss": compute_miou_loss(pred["seg"], feed["labels"], per_instance=False)}
        labels = torch.max(pred["seg"], dim=-2)[1] if test else feed["labels"]
        nocs_loss = compute_nocs_loss(pred["nocs"], feed["nocs"], labels=labels, confidence=None, loss="l2",
                                      self_supervise=False, per_instance=False, sym=self.sym, pwm_num=self.pwm_num)
        if self.sym:
            loss_dict["nocs_dist_loss"], loss_dict["nocs_pwm_loss"] = nocs_loss
        else:
            loss_dict["nocs_loss"] = nocs_loss
        self._pose_terms(feed["gt_part"], pred["part"], feed["init_part"], loss_dict)
        loss_dict["corner_loss"] = compute_point_pose_loss(feed["gt_part"], pred["part"], self._gt_box(feed["meta"]),
                                                           metric=self.pose_loss_type["point"])[0]
        self.summarize_losses(loss_dict)

    def test(self, save=False, no_eval=False, epoch=0):
        self.prepare_data()
        self.loss_dict = {}
        with torch.no_grad():
            self.pred_dict = self.net(self.feed_dict, test=True)
            self.pred_dict.pop("_labels_i32", None)      # (the track loop's private hand-over: not part of the reference's dict)
            if not no_eval:
                self.compute_loss(test=True)

    def update(self):
        self.prepare_data()
        self.pred_dict = self.net(self.feed_dict)
        self.compute_loss()
        self.loss_dict["total_loss"].backward()


class RotationModel(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.net = PartCanonNet(cfg)
        self.raw_feed_dict = {}

    def set_data(self, data):
        self.raw_feed_dict = data

    def prepare_data(self, data):
        gt_part, init_part = self.prepare_poses(data)
        feed = cvt_torch({"points": data["points"], "points_mean": data["meta"]["points_mean"], "nocs": data["nocs"],
                          "state": {"part": init_part}, "gt_part": gt_part}, self.device)
        feed["meta"] = data["meta"]
        feed["labels"] = data["labels"].long().to(self.device)
        part = feed["state"]["part"]
        canon = {k: part[k].reshape((-1,) + part[k].shape[2:]) for k in ("rotation", "translation", "scale")}   # (B*P, ...)
        feed["canon_pose"] = canon
        B = len(feed["gt_part"]["scale"])
        feed["root_delta"] = compute_parts_delta_pose(part, feed["gt_part"],
                                                      {k: v.reshape((B, self.num_parts) + v.shape[1:]) for k, v in canon.items()})
        self.feed_dict = feed

    def compute_loss(self, test_mode=False, per_instance=False):
        feed, pred = self.feed_dict, self.pred_dict
        loss_dict = {}
        per, init_per = self._pose_terms(feed["gt_part"], pred["part"], feed["state"]["part"], loss_dict, per_instance=per_instance)
        if per_instance:      # test(save=True): the per-instance table the harness dumps as CSV (reference model.py:264-266)
            per.update({f"init_{k}": v for k, v in init_per.items()})
            self.record_per_diff(feed, {"test": per})
        loss_dict["corner_loss"] = compute_point_pose_loss(feed["gt_part"], pred["part"], self._gt_box(feed["meta"]),
                                                           metric=self.pose_loss_type["point"])[0]
        if "point_rotation" in pred:
            labels = feed["labels"]
            onehot = torch.cat([torch.eye(self.num_parts), torch.zeros(2, self.num_parts)], dim=0).to(labels.device)
            part_mask = onehot[labels].transpose(-1, -2)                                  # (B,P,N)
            point_rot = pred["point_rotation"]                                             # (B,P,N,3,3)
            gt_rot = feed["root_delta"]["rotation"].unsqueeze(-3)                          # (B,P,1,3,3)
            if point_rot.shape[1] == 1 and point_rot.dim() > gt_rot.dim():
                point_rot = point_rot.squeeze(1)
            rloss = rot_yaxis_loss(gt_rot, point_rot) if self.sym else rot_trace_loss(gt_rot, point_rot, metric=self.pose_loss_type["r"])
            loss_dict["rloss"] = torch.sum(rloss * part_mask) / torch.clamp(torch.sum(part_mask), min=1.0)
        self.summarize_losses(loss_dict)

    def update(self):
        self.prepare_data(self.raw_feed_dict)
        self.pred_dict = self.net(self.feed_dict, test_mode=False)
        self.compute_loss(test_mode=False)
        self.loss_dict["total_loss"].backward()

    def test(self, save=False, no_eval=False, epoch=0):
        with torch.no_grad():
            self.prepare_data(self.raw_feed_dict)
            self.pred_dict = self.net(self.feed_dict, test_mode=True)
            self.compute_loss(test_mode=True, per_instance=save)
