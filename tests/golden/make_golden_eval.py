"""Golden fixture G10 for the evaluation tables: the reference's own
pose_utils/bbox_utils.py::eval_single_part_iou and misc/eval/eval.py::get_joint_state run on seeded inputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py [--ref /root/reference]

Only inputs (regenerated from the seed by the test) and OUTPUT arrays are stored; nothing of the reference is copied.
"""
from __future__ import annotations

import argparse
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from tests import clouds  # noqa: E402


def make_inputs(seed: int, P: int):
    """Seeded (gt corners, pred corners, gt pose, pred pose) for one frame; shared with tests/test_eval_cpu.py."""
    rng = np.random.default_rng(seed)
    half = 0.1 + 0.3 * rng.random((P, 3))
    gt_corners = np.stack([-half, half], axis=1).astype(np.float32)                       # (P,2,3)
    pred_corners = (gt_corners * (1.0 + 0.1 * rng.standard_normal((P, 2, 3)))).astype(np.float32)
    def pose(jitter):
        rot = np.stack([clouds._rot_y(0.7 * p + jitter * rng.standard_normal()) @ clouds._rot_x(0.3 + jitter * rng.standard_normal())
                        for p in range(P)]).astype(np.float32)
        trans = (np.array([0.1, -0.05, 1.0]) + 0.2 * np.arange(P)[:, None] + jitter * rng.standard_normal((P, 3))).astype(np.float32)
        scale = (0.3 + 0.02 * np.arange(P) + 0.2 * jitter * rng.standard_normal(P)).astype(np.float32)
        return {"rotation": rot, "translation": trans[..., None], "scale": scale}
    gt = pose(0.0)
    pred = pose(0.05)
    return gt_corners, pred_corners, gt, pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    for name in ("cv2", "trimesh"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for p in (a.ref, f"{a.ref}/network", f"{a.ref}/pose_utils", f"{a.ref}/misc/eval"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pose_utils.bbox_utils import eval_single_part_iou  # noqa
    from pose_utils.metrics import rot_diff_degree  # noqa

    out = {}
    for tag, P, sym, nocs in (("rigid_sym", 1, True, True), ("rigid", 1, False, True), ("arti", 4, False, False)):
        gc, pc, gt, pred = make_inputs(11 + P + int(sym), P)
        t = lambda d: {k: torch.from_numpy(v).unsqueeze(0) for k, v in d.items()}
        avg, per = eval_single_part_iou(torch.from_numpy(gc).unsqueeze(0), torch.from_numpy(pc).unsqueeze(0), t(gt), t(pred),
                                        separate="both", nocs=nocs, sym=sym)
        for name in ("npcs_iou", "iou", "gt_bbox_iou"):
            out[f"{tag}_{name}"] = np.array([float(per[name][p][0]) for p in range(P)], np.float64)
    # joint states of the articulated case: the reference's own misc/eval/eval.py::get_joint_state (prismatic drawers)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_eval", f"{a.ref}/misc/eval/eval.py")
    ref_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_eval)
    gc, pc, gt, pred = make_inputs(11 + 4, 4)
    info = {"tree": [3, 3, 3, -1], "type": "prismatic", "main_axis": [2, 2, 2]}
    out["arti_joint_state_gt"] = np.asarray(ref_eval.get_joint_state(info, gt), np.float64)
    out["arti_joint_state_pred"] = np.asarray(ref_eval.get_joint_state(info, pred), np.float64)
    out["rot_diff_deg"] = rot_diff_degree(torch.from_numpy(gt["rotation"]), torch.from_numpy(pred["rotation"])).numpy()
    np.savez_compressed(HERE / "g10_eval.npz", **out)
    print({k: v for k, v in out.items()})


if __name__ == "__main__":
    main()
