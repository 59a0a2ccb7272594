"""Configuration: three YAML trees merged into one dict, any flag named `a/b` overrides
cfg['a']['b'] (mirrors the reference's configs/config.py:8-71; same resulting keys)."""
from __future__ import annotations

import copy
import os
from os.path import join as pjoin

import torch
import yaml

_BASE = os.path.dirname(__file__)


def _override(cfg: dict, dotted: str, path: list, value) -> None:
    head = path[0]
    if len(path) == 1:
        if cfg.get(head) != value:
            cfg[head] = value
        return
    cfg.setdefault(head, {})
    _override(cfg[head], dotted, path[1:], value)


def _load(rel: str):
    with open(pjoin(_BASE, rel), "r") as f:
        return yaml.safe_load(f)


def _finish(cfg: dict, make_dirs: bool) -> dict:
    obj_cfg = _load(pjoin("obj_config", cfg["obj_config"]))
    cfg["pointnet"] = {k: _load(pjoin("pointnet_config", v)) for k, v in cfg["pointnet_cfg"].items()}
    if make_dirs:
        os.makedirs(cfg["experiment_dir"], exist_ok=True)
    cfg["num_expr"] = cfg["experiment_dir"].split("/")[-1]
    cat = str(cfg["obj_category"])
    cfg["obj_category"] = cat
    cfg["num_parts"] = obj_cfg[cat]["num_parts"]
    cfg["num_joints"] = obj_cfg[cat]["num_joints"]
    cfg["obj_tree"] = obj_cfg[cat]["tree"]
    cfg["obj_sym"] = obj_cfg[cat]["sym"]
    cfg["obj"] = obj_cfg
    cfg["obj_info"] = obj_cfg[cat]
    cfg["root_dset"] = obj_cfg["basepath"]
    cfg["device"] = torch.device("cuda:%d" % cfg["cuda_id"]) if torch.cuda.is_available() else "cpu"
    return cfg


def get_config(args, save: bool = True) -> dict:
    """args: argparse.Namespace with `config` plus optional `a/b`-named overrides."""
    cfg = _load(pjoin("all_config", args.config))
    overrides = dict(vars(args))
    overrides.pop("config")
    for key, item in overrides.items():
        if item is not None:
            _override(cfg, key, key.split("/"), item)
    cfg = _finish(cfg, make_dirs=True)
    if save:
        with open(pjoin(cfg["experiment_dir"], "config.yml"), "w") as f:
            yaml.safe_dump({k: v for k, v in cfg.items() if k not in ("device", "obj", "pointnet")}, f)
    return cfg


def make_config(obj_category="1", obj_config="obj_info_nocs.yml", config="config_track.yml", **overrides) -> dict:
    """Programmatic equivalent of get_config for tests / bench: no argparse, no directories."""
    cfg = copy.deepcopy(_load(pjoin("all_config", config)))
    cfg["obj_category"] = str(obj_category)
    cfg["obj_config"] = obj_config
    for key, item in overrides.items():
        _override(cfg, key, key.split("/"), item)
    return _finish(cfg, make_dirs=False)
