"""Golden fixture G13: the loss dict of the reference's `EvalTrackModel.test(no_eval=False)` — pose errors of the
prediction and of its initialisation, CoordinateNet's segmentation / NOCS losses, the three box IoUs — on the seeded
trajectories of G9 (bottle: rigid, symmetric — best of 20 rotations of the ground truth about y; camera: rigid,
non-symmetric; drawers: 4 parts; oriented-box IoU by occupancy on a 50^3 grid throughout).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trackloss.py [--ref /root/reference]

Same run-time adjustments as make_golden.py's G9 (FPS start 0, CUDA-semantics three_nn / ball query, seeds 1234).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from tests import clouds  # noqa: E402
from tests.golden.make_golden import import_reference, ref_cfg  # noqa: E402
from tests.golden.make_golden_train import ForceFpsStartZero, cuda_semantics  # noqa: E402
from tests.weights import make_state_dict  # noqa: E402

CASES = [("bottle", "1", "obj_info_nocs.yml", "nocs", 3), ("camera", "3", "obj_info_nocs.yml", "nocs", 3),
         ("drawers", "drawers", "obj_info_sapien.yml", "arti", 3)]


def flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        key = f"{prefix}{k}"
        if isinstance(v, dict):
            out.update(flatten(v, key + "/"))
        else:
            out[key] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v, np.float64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    pu = import_reference(args.ref)
    assert not pu.CUDA
    cuda_semantics(pu)
    torch.set_num_threads(8)
    from trainer import Trainer
    out = {}
    for tag, cat, objcfg, kind, frames in CASES:
        cfg = ref_cfg(args.ref, cat, objcfg)
        cfg["init_frame"]["gt"] = False
        cfg["track_cfg"]["gt_label"] = (tag == "drawers")
        with contextlib.redirect_stdout(io.StringIO()):
            trainer = Trainer(cfg)
        model = trainer.model.eval()
        model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=7))
        data = clouds.make_trajectory(kind, 2, frames, seed=0)
        torch.manual_seed(1234)
        np.random.seed(1234)
        with ForceFpsStartZero():
            model.set_data(data)
            model.test(save=False, no_eval=False)
        flat = flatten({k: v for k, v in model.loss_dict.items() if not k.startswith("frame_")})
        for k, v in flat.items():
            out[f"{tag}/{k}"] = v
        print(tag, {k: round(float(v), 5) for k, v in flat.items() if "iou" in k or k in ("avg_seg", "avg_nocs")}, flush=True)
    np.savez_compressed(HERE / "g13_trackloss.npz", **out)
    print("wrote", HERE / "g13_trackloss.npz")


if __name__ == "__main__":
    main()
