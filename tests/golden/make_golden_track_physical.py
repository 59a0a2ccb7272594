"""Golden G9p: free-running EvalTrackModel trajectories of the REFERENCE under physical-regime weights.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_track_physical.py [--ref /root/reference]

G9 (make_golden.py) runs the reference's track loop with purely random weights: its pose fit then returns
unphysical poses (negative scales on the drawers fixture), the loop amplifies rounding noise and only a
teacher-forced comparison can hold 1e-4.  This fixture uses `tests/weights.py::make_physical_state_dict`
(random SA / FP stack + a planted coordinate pass-through and near-identity rotation heads) so that the
reference's own loop tracks: every frame of every trajectory is a 1e-4 target for a free-running run.

Same method as make_golden.py: the reference is imported read-only (CPU path, CUDA neighbour semantics
patched in, FPS start forced to 0, seeds fixed); only OUTPUT poses are written (tests/golden/g9p_track.npz),
plus the smallest segmentation-logit gap and the per-frame label counts the generator asserted on.
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
sys.path.insert(0, str(HERE))

from make_golden import (ForceFpsStartZero, import_reference, query_ball_point_cuda_semantics, ref_cfg,  # noqa: E402
                         three_nn_cuda_semantics)
from tests import clouds  # noqa: E402
from tests.weights import make_physical_state_dict  # noqa: E402

SETUPS = clouds.PHYSICAL_SETUPS      # one table for the generator and the tests


def patch_cuda_semantics(pu):
    """The reference's CPU neighbour searches test expanded-form distances; the network goldens use the CUDA kernels'
    semantics (direct form, strict '<', sqrt of the squared 3-NN distance) -- the very patches of make_golden.py."""
    pu.three_nn = three_nn_cuda_semantics
    pu.query_ball_point = query_ball_point_cuda_semantics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--set", default="first", choices=["first", "more"],
                    help="first: clouds.PHYSICAL_SETUPS -> g9p_track.npz; more: clouds.PHYSICAL_SETUPS_MORE -> g9p_track_more.npz")
    ap.add_argument("--scan", nargs=2, metavar=("TAG", "WSEEDS"), default=None,
                    help="seed search, nothing written: run TAG of the chosen set with each weight seed of the comma list and print its margins")
    args = ap.parse_args()
    setups, target = (SETUPS, "g9p_track.npz") if args.set == "first" else (clouds.PHYSICAL_SETUPS_MORE, "g9p_track_more.npz")
    pu = import_reference(args.ref)
    assert not pu.CUDA
    patch_cuda_semantics(pu)
    torch.set_num_threads(8)
    from trainer import Trainer
    out = {}
    if args.scan:
        base = setups[args.scan[0]]
        setups = {f"{args.scan[0]}@{w}": base[:5] + (int(w),) + base[6:] for w in args.scan[1].split(",")}
    for tag, (cat, objcfg, kind, frames, batch, wseed, tseed) in setups.items():
        cfg = ref_cfg(args.ref, cat, objcfg)
        cfg["init_frame"]["gt"] = False
        with contextlib.redirect_stdout(io.StringIO()):
            trainer = Trainer(cfg)
        model = trainer.model.eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(make_physical_state_dict(shapes, wseed, cfg["num_parts"], bool(cfg["obj_sym"]), kind))
        data = clouds.make_trajectory(kind, batch, frames, seed=7)
        torch.manual_seed(tseed)
        np.random.seed(tseed)
        with ForceFpsStartZero():
            model.set_data(data)
            model.test(save=False, no_eval=True)
        poses = model.pred_dict["poses"]
        gap_min, counts = np.inf, []
        for i, pose in enumerate(poses):
            for key in ("rotation", "translation", "scale"):
                out[f"{tag}_{i}_{key}"] = pose[key].numpy()
            if i == 0:
                continue
            seg = model.pred_dict["npcs_pred"][i]["seg"]                      # (B,P+e,N) softmax
            top2 = torch.topk(seg, 2, dim=1)[0]
            gap_min = min(gap_min, float((top2[:, 0] - top2[:, 1]).min()))
            lab = torch.argmax(seg, dim=1)
            counts.append([[int((lab[b] == p).sum()) for p in range(cfg["num_parts"])] for b in range(batch)])
        scales = np.stack([out[f"{tag}_{i}_scale"] for i in range(len(poses))])
        print(f"{tag}: scale range [{scales.min():.4f}, {scales.max():.4f}], min softmax gap {gap_min:.2e}, "
              f"label counts per part min {np.min(counts)}")
        if args.scan:
            continue
        assert scales.min() > 0.05 and scales.max() < 2.0, "trajectory left the physical regime"
        assert gap_min > 2e-5, f"{tag}: a point sits on a segmentation decision boundary (gap {gap_min:.2e}): pick another seed"
        assert np.min(counts) > 16, "a part is left with a handful of points: pick another seed"
        out[f"{tag}_min_softmax_gap"] = np.float32(gap_min)
        out[f"{tag}_label_counts"] = np.asarray(counts, np.int32)
    if args.scan:
        return
    np.savez_compressed(HERE / target, **out)
    print("wrote", HERE / target)


if __name__ == "__main__":
    main()
