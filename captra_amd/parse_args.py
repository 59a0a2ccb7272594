"""The command-line surface shared by the harnesses: every flag of the reference's network/parse_args.py:5-69, same
names, types and defaults (a flag named `a/b` overrides cfg['a']['b'], None = keep the YAML value).  Declared as a table
so that `track`, `train` and `eval` register exactly the same set; flags whose subsystem is outside this build
(`--num_workers`, `--dataset_length`, `--eval_train`: DataLoader options) are accepted and ignored.
"""
from __future__ import annotations

import argparse


def boolean_string(s: str) -> bool:
    if s not in ("True", "False"):
        raise ValueError("Not a valid boolean string")
    return s == "True"


_FLAGS = [  # (name, type, default)
    ("obj_config", str, None), ("obj_category", str, None), ("experiment_dir", str, None), ("resume_epoch", int, -1),
    ("coord_exp/dir", str, None), ("coord_exp/resume_epoch", int, None),
    ("batch_size", int, None), ("cuda_id", int, None), ("num_points", int, None), ("data_radius", float, None),
    ("num_workers", int, 0), ("dataset_length", int, None), ("pointnet_cfg/camera", str, None),
    ("network/type", str, None), ("network/nocs_head_dims", int, None), ("network/backbone_out_dim", int, None),
    ("network/pwm_num", int, None),
    ("init_frame/gt", boolean_string, None), ("nocs_otf", boolean_string, None),
    ("track_cfg/gt_label", boolean_string, None), ("track_cfg/nocs2d_label", boolean_string, None), ("track_cfg/nocs2d_path", str, None),
    # optimisation
    ("total_epoch", int, None), ("optimizer", str, None), ("weight_decay", float, None), ("learning_rate", float, None),
    ("lr_policy", str, None), ("lr_gamma", float, None), ("lr_step_size", int, None), ("lr_clip", float, None), ("freq/save", int, None),
]
_FLAGS += [(f"loss_weight/{k}", float, None) for k in ("rloss", "tloss", "sloss", "corner_loss", "nocs_loss", "nocs_dist_loss",
                                                        "nocs_pwm_loss", "seg_loss")]
_FLAGS += [(f"pose_loss_type/{k}", str, None) for k in ("r", "s", "t", "point")]
_FLAGS += [("pose_perturb/type", str, None)] + [(f"pose_perturb/{k}", float, None) for k in ("r", "s", "t")]
_SWITCHES = ("save", "eval_train", "no_eval")
HARNESS_ONLY = ("num_workers", "dataset_length", "eval_train", "save", "no_eval")     # not configuration overrides


def add_args(parser: argparse.ArgumentParser, default_config: str = "config.yml") -> argparse.ArgumentParser:
    parser.add_argument("--config", type=str, default=default_config)
    for name, typ, default in _FLAGS:
        parser.add_argument(f"--{name}", type=typ, default=default)
    for name in _SWITCHES:
        parser.add_argument(f"--{name}", action="store_true", default=False)
    return parser
