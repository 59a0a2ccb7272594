"""`Trainer`: model factory, checkpoint resume and the `test(data)` entry of the tracking harness.

Mirrors the inference surface of the reference's network/trainer.py: `Trainer(cfg, logger)`
(l.86-118), `.resume()` (l.147-194, incl. loading the CoordNet experiment's weights under
`npcs_net.*`, l.159-169), `.save()` (l.196-210) and `.test(data, save, no_eval)` (l.223-230).
Training (`update`, optimiser, LR / BN-momentum schedules, l.120-145, 212-221) is out of scope
for the hot path.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.nn as nn

from .model import EvalTrackModel
from .utils import ensure_dirs


def get_last_model(dirname, key=""):
    if not dirname or not os.path.exists(dirname):
        return None
    models = sorted(pjoin(dirname, f) for f in os.listdir(dirname)
                    if os.path.isfile(pjoin(dirname, f)) and key in f and f.endswith(".pt"))
    return models[-1] if models else None


class Trainer(nn.Module):
    def __init__(self, cfg, logger=None):
        super().__init__()
        self.ckpt_dir = pjoin(cfg["experiment_dir"], "ckpt")
        self.device = cfg["device"]
        self.network_type = cfg["network"]["type"]
        if self.network_type != "rot_coord_track":
            raise NotImplementedError(f"network.type={self.network_type}: only the tracking model "
                                      "(rot_coord_track / EvalTrackModel) is on the hot path")
        self.model = EvalTrackModel(cfg)
        coord = cfg.get("coord_exp") or {}
        self.coord_exp_dir = pjoin(coord["dir"], "ckpt") if coord.get("dir") else None
        self.coord_resume_epoch = coord.get("resume_epoch", -1)
        self.optimizer = None
        self.epoch = 0
        self.iteration = 0
        self.loss_dict = {}
        self.cfg = cfg
        self.logger = logger
        self.to(self.device)

    def log_string(self, msg):
        print(msg)
        if self.logger is not None:
            self.logger.info(msg)

    @staticmethod
    def _pick(dirname, resume_epoch):
        name = get_last_model(dirname)
        if resume_epoch is not None and resume_epoch > 0:
            cand = pjoin(dirname, f"model_{resume_epoch:04d}.pt")
            if os.path.exists(cand):
                name = cand
        return name

    def resume(self):
        ckpt = OrderedDict()
        if self.coord_exp_dir is not None:
            coord_name = self._pick(self.coord_exp_dir, self.coord_resume_epoch)
            if coord_name is None:
                raise FileNotFoundError(f"no CoordNet checkpoint under {self.coord_exp_dir}")
            self.log_string(f"Load CoordNet model from {coord_name}")
            coord_state = torch.load(coord_name, map_location=self.device)["model"]
            for key, value in coord_state.items():
                if key.startswith("net"):
                    ckpt["npcs_net" + key[3:]] = value
        model_name = self._pick(self.ckpt_dir, self.cfg.get("resume_epoch", -1))
        if model_name is None:
            self.log_string("Initialize from 0")
        else:
            state = torch.load(model_name, map_location=self.device)
            self.epoch, self.iteration = state["epoch"], state["iteration"]
            ckpt.update(state["model"])
            self.log_string("Resume from epoch %d" % self.epoch)
        self.model.load_state_dict(ckpt, strict=False)
        return self.epoch

    def save(self, name=None, extra_info=None):
        ensure_dirs(self.ckpt_dir)
        name = name or f"model_{self.epoch:04d}"
        path = pjoin(self.ckpt_dir, f"{name}.pt")
        state = {"epoch": self.epoch, "iteration": self.iteration, "model": self.model.state_dict(), "optimizer": {}}
        if isinstance(extra_info, dict):
            state.update(extra_info)
        torch.save(state, path)
        self.log_string(f"Saving model at epoch {self.epoch}, path {path}")

    def test(self, data, save=False, no_eval=False):
        self.model.eval()
        self.model.set_data(data)
        self.model.test(save=save, no_eval=no_eval, epoch=self.epoch)
        return self.model.pred_dict, self.model.loss_dict
