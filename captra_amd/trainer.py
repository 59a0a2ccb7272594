"""`Trainer`: model factory, checkpoint resume, the `test(data)` entry of the tracking harness and the training step.

Mirrors the reference's network/trainer.py: `Trainer(cfg, logger)` (l.86-118) for the three experiment types
(`canon_coord` -> CanonCoordModel, `rot` -> RotationModel, `rot_coord_track` -> EvalTrackModel), `.resume()` (l.147-194,
incl. loading the CoordNet experiment's weights under `npcs_net.*`, l.159-169), `.save()` (l.196-210),
`.test(data, save, no_eval)` (l.223-230), and for training `weights_init` (l.18-40), the Adam / SGD
optimiser and StepLR schedule (l.43-74), `.step_epoch()` (LR clip + BatchNorm momentum decay, l.125-145) and
`.update(data)` (l.212-221).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.nn as nn
import torch.nn.init as init
from torch.optim import lr_scheduler

from .model import EvalTrackModel
from .utils import ensure_dirs


def weights_init(init_type="gaussian"):
    """Initialiser applied to every Conv* / Linear* module by class-name prefix; biases to zero."""
    def fn(m):
        name = m.__class__.__name__
        if not ((name.startswith("Conv") or name.startswith("Linear")) and hasattr(m, "weight")):
            return
        if init_type == "gaussian":
            init.normal_(m.weight.data, 0.0, 0.02)
        elif init_type == "xavier":
            init.xavier_normal_(m.weight.data, gain=math.sqrt(2))
        elif init_type == "kaiming":
            init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
        elif init_type == "orthogonal":
            init.orthogonal_(m.weight.data, gain=math.sqrt(2))
        elif init_type != "default":
            raise ValueError(f"unsupported initialization {init_type}")
        if getattr(m, "bias", None) is not None:
            init.constant_(m.bias.data, 0.0)
    return fn


def get_optimizer(params, cfg):
    if len(params) == 0 or "optimizer" not in cfg:
        return None
    if cfg["optimizer"] == "Adam":
        return torch.optim.Adam(params, lr=cfg["learning_rate"], betas=(0.9, 0.999), eps=1e-08, weight_decay=cfg["weight_decay"])
    if cfg["optimizer"] == "SGD":
        return torch.optim.SGD(params, lr=cfg["learning_rate"], momentum=0.9)
    raise ValueError(f"unsupported optimizer {cfg['optimizer']}")


def get_scheduler(optimizer, cfg, it=-1):
    if optimizer is None or cfg.get("lr_policy", "constant") == "constant":
        return None
    if cfg["lr_policy"] == "step":
        return lr_scheduler.StepLR(optimizer, step_size=cfg["lr_step_size"], gamma=cfg["lr_gamma"], last_epoch=it)
    raise ValueError(f"lr_policy {cfg['lr_policy']} not implemented")


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
        if self.network_type == "rot_coord_track":
            self.model = EvalTrackModel(cfg)
        elif self.network_type in ("canon_coord", "rot"):
            from .train_models import CanonCoordModel, RotationModel
            self.model = CanonCoordModel(cfg) if self.network_type == "canon_coord" else RotationModel(cfg)
        else:
            raise NotImplementedError(f"network.type={self.network_type}")
        coord = cfg.get("coord_exp") or {}
        self.coord_exp_dir = pjoin(coord["dir"], "ckpt") if (self.network_type == "rot_coord_track" and coord.get("dir")) else None
        self.coord_resume_epoch = coord.get("resume_epoch", -1)
        self.optimizer = get_optimizer([p for p in self.model.parameters() if p.requires_grad], cfg)
        self.scheduler = get_scheduler(self.optimizer, cfg)
        if "weight_init" in cfg:
            self.apply(weights_init(cfg["weight_init"]))
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
            if self.optimizer is not None and state.get("optimizer"):
                try:      # a checkpoint whose parameter groups do not match (added / missing parameters) keeps the
                    self.optimizer.load_state_dict(state["optimizer"])   # weights and drops the optimiser state (trainer.py:181-186)
                except (ValueError, KeyError):
                    self.log_string("Optimizer state of the checkpoint does not match this model: discarded")
                self.scheduler = get_scheduler(self.optimizer, self.cfg, self.epoch)
        self.model.load_state_dict(ckpt, strict=False)
        return self.epoch

    def step_epoch(self):
        """Start of an epoch: StepLR until the rate falls to lr_clip, BatchNorm momentum decayed every momentum_step_size."""
        cfg = self.cfg
        self.epoch += 1
        if self.scheduler is not None and self.scheduler.get_last_lr()[0] > cfg["lr_clip"]:
            self.scheduler.step()
        if self.scheduler is not None:
            self.lr = self.scheduler.get_last_lr()[0]
            self.log_string("Epoch %d/%d, learning rate = %f" % (self.epoch, cfg["total_epoch"], self.lr))
        momentum = max(cfg["momentum_original"] * (cfg["momentum_decay"] ** (self.epoch // cfg["momentum_step_size"])),
                       cfg["momentum_min"])
        self.log_string("BN momentum updated to %f" % momentum)
        self.momentum = momentum
        for m in self.model.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.momentum = momentum

    def update(self, data):
        """One optimisation step on a batch (reference trainer.py:212-221): forward in training mode, the experiment's
        losses, backward, optimiser step.  Returns the loss dict."""
        self.optimizer.zero_grad()
        self.model.train()
        self.model.set_data(data)
        self.model.update()
        loss_dict = self.model.loss_dict
        self.loss_dict.update(loss_dict)
        if torch.distributed.is_available() and torch.distributed.is_initialized():     # one process per GPU: average gradients
            from .parallel import allreduce_gradients
            allreduce_gradients([p for p in self.model.parameters() if p.requires_grad], torch.distributed.get_world_size())
        self.optimizer.step()
        self.iteration += 1
        return loss_dict

    def save(self, name=None, extra_info=None):
        ensure_dirs(self.ckpt_dir)
        name = name or f"model_{self.epoch:04d}"
        path = pjoin(self.ckpt_dir, f"{name}.pt")
        state = {"epoch": self.epoch, "iteration": self.iteration, "model": self.model.state_dict(),
                 "optimizer": self.optimizer.state_dict() if self.optimizer is not None else {}}
        if isinstance(extra_info, dict):
            state.update(extra_info)
        torch.save(state, path)
        self.log_string(f"Saving model at epoch {self.epoch}, path {path}")

    def test(self, data, save=False, no_eval=False):
        self.model.eval()
        self.model.set_data(data)
        self.model.test(save=save, no_eval=no_eval, epoch=self.epoch)
        return self.model.pred_dict, self.model.loss_dict
