"""`python -m captra_amd.eval`: error tables from the per-trajectory result pickles (counterpart of the reference's
misc/eval/eval.py:27-111).  For every frame after the first of every `<experiment_dir>/results/data/*.pkl`
(written by `captra_amd.track --save`, layout of model.py:482-509): rotation / translation / scale errors and the
5 deg 5 cm / 10 deg 10 cm flags per part (part_dof_utils.py:54-67), box IoUs (pose_utils/bbox_utils.py), and for
articulated objects the joint-state error; writes results/err.pkl + results/err.csv and prints the averages.
"""
from __future__ import annotations

import argparse
import os
import pickle
from os.path import join as pjoin

import numpy as np
import torch

from .configs.config import get_config
from .pose_utils.bbox_utils import eval_instance_part_iou
from .pose_utils.metrics import rot_diff_degree
from .pose_utils.part_dof_utils import eval_part_full


def get_joint_state(info: dict, pose: dict) -> np.ndarray:
    """Joint value of every child part w.r.t. its parent: relative rotation angle (revolute) or the offset along the
    joint's main axis in the parent frame (prismatic); eval.py:58-77."""
    states = []
    for child, parent in enumerate(info["tree"]):
        if parent == -1:
            continue
        rot, trans = np.asarray(pose["rotation"], np.float64), np.asarray(pose["translation"], np.float64)
        if info["type"] == "revolute":
            states.append(float(rot_diff_degree(torch.as_tensor(pose["rotation"][child]), torch.as_tensor(pose["rotation"][parent]))))
        else:
            rel = rot[parent].T @ (trans[child] - trans[parent])
            states.append(float(rel.reshape(-1)[info["main_axis"][len(states)]]))
    return np.array(states)


def eval_data(name: str, data: dict, obj_info: dict) -> dict:
    sym, rigid = obj_info["sym"], obj_info["num_parts"] == 1
    gt_corners = np.asarray(data["gt"]["corners"])
    errors = {}
    for i in range(1, len(data["pred"]["poses"])):          # frame 0 is the initialisation
        gt = {k: torch.as_tensor(np.asarray(v)) for k, v in data["gt"]["poses"][i].items()}
        pred = {k: torch.as_tensor(np.asarray(v)) for k, v in data["pred"]["poses"][i].items()}
        _, per = eval_part_full(gt, pred, per_instance=True, yaxis_only=sym)
        row = {k: float(np.asarray(v)) for k, v in per.items()}
        iou = eval_instance_part_iou(gt_corners, np.asarray(data["pred"]["corners"][i]), {k: v.numpy() for k, v in gt.items()},
                                   {k: v.numpy() for k, v in pred.items()}, nocs=rigid, sym=sym)
        row.update({f"iou_{j}": float(v) for j, v in enumerate(iou["iou"])})
        if not rigid:
            diff = np.abs(get_joint_state(obj_info, {k: v.numpy() for k, v in pred.items()})
                          - get_joint_state(obj_info, {k: v.numpy() for k, v in gt.items()}))
            row.update({f"theta_diff_{j}": float(v) for j, v in enumerate(diff)})
        errors[f"{name}_{i}"] = row
    return errors


def write_csv(errors: dict, path: str) -> None:
    keys = list(next(iter(errors.values())).keys())
    with open(path, "w") as f:
        f.write("," + ",".join(keys) + "\n")
        for inst, row in errors.items():
            f.write(inst + "," + ",".join(str(row[k]) for k in keys) + "\n")


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", type=str, default="config_track.yml")
    ap.add_argument("--obj_config", type=str, default=None)
    ap.add_argument("--obj_category", type=str, default=None)
    ap.add_argument("--experiment_dir", type=str, default=None)
    cfg = get_config(ap.parse_args(argv), save=False)
    data_path = pjoin(cfg["experiment_dir"], "results", "data")
    errors = {}
    for raw in sorted(os.listdir(data_path)):
        with open(pjoin(data_path, raw), "rb") as f:
            errors.update(eval_data(raw.rsplit(".", 1)[0], pickle.load(f), cfg["obj_info"]))
    if not errors:
        raise SystemExit(f"no result pickles under {data_path}")
    err_path = pjoin(cfg["experiment_dir"], "results", "err.pkl")
    with open(err_path, "wb") as f:
        pickle.dump(errors, f)
    write_csv(errors, err_path.replace("pkl", "csv"))
    avg = {k: float(np.mean([row[k] for row in errors.values()])) for k in next(iter(errors.values()))}
    for k, v in avg.items():
        print(f"{k}: {v}")
    return avg


if __name__ == "__main__":
    main()
