"""Golden G15: the REFERENCE's track loop with the on-the-fly re-crop (`nocs_otf=True`, network/models/model.py:425-452 ->
datasets/nocs_data/nocs_data_process.py:182-236 full_data_from_depth_image with `pre_fetched`, no cv2 call on that path)
under physical-regime weights, B = 1 (the reference asserts batch 1 for this mode, model.py:319).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_otf_loop.py [--ref /root/reference]

Per trajectory: frames of captra_amd.synthetic.make_otf_trajectory (a blob drifting over a synthetic depth image), seeded
weights (make_physical_state_dict), the seeded perturbed initial pose.  Written: every `pred_dict['poses'][i]` and the
re-cropped, mean-subtracted cloud + labels the loop fed to the networks at every frame.

Run-time adjustments (monkey patches, never edits; the same ones as make_golden_otf.py / make_golden_track_physical.py):
the data-side `farthest_point_sample` takes its GPU branch (thin to 5 x npoint with numpy.random.permutation, then FPS) with
the CUDA FPS replaced by the CPU oracle FPS (start 0, pinned by G1); the network-side FPS start forced to 0; CUDA neighbour
semantics for ball query / three_nn; torch / numpy generators seeded.
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

from make_golden import ForceFpsStartZero, import_reference, ref_cfg  # noqa: E402
from make_golden_track_physical import patch_cuda_semantics  # noqa: E402
from captra_amd.synthetic import OTF_LOOP_SETUPS as SETUPS  # noqa: E402  (one table for the generator and the tests)
from captra_amd.synthetic import make_otf_trajectory, make_physical_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    pu = import_reference(args.ref)
    assert not pu.CUDA
    patch_cuda_semantics(pu)
    torch.set_num_threads(8)
    from oracle import ops as O
    from trainer import Trainer
    import model as ref_model

    def fps_gpu_branch(xyz, npoint, device):
        if len(xyz) > 5 * npoint:
            idx = np.random.permutation(len(xyz))[:5 * npoint]
            return idx[O.furthest_point_sample(np.asarray(xyz[idx], np.float32)[None], npoint)[0]]
        return O.furthest_point_sample(np.asarray(xyz, np.float32)[None], npoint)[0]

    # the module object model.py's `full_data_from_depth_image` lives in (imported there as datasets.nocs_data.nocs_data_process)
    ref_model.full_data_from_depth_image.__globals__["farthest_point_sample"] = fps_gpu_branch

    out = {}
    for tag, (frames, dseed, wseed, tseed) in SETUPS.items():
        cfg = ref_cfg(args.ref, "1", "obj_info_nocs.yml")
        cfg["init_frame"]["gt"] = False
        cfg["nocs_otf"] = True
        cfg["batch_size"] = 1
        with contextlib.redirect_stdout(io.StringIO()):
            trainer = Trainer(cfg)
        model = trainer.model.eval()
        assert model.nocs_otf
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(make_physical_state_dict(shapes, wseed, cfg["num_parts"], bool(cfg["obj_sym"]), "nocs"))
        data = make_otf_trajectory(1, frames, seed=dseed)
        torch.manual_seed(tseed)
        np.random.seed(tseed)
        with ForceFpsStartZero():
            model.set_data(data)
            model.test(save=False, no_eval=True)
        poses = model.pred_dict["poses"]
        for i, pose in enumerate(poses):
            for key in ("rotation", "translation", "scale"):
                out[f"{tag}_{i}_{key}"] = pose[key].numpy()
            if i > 0:
                out[f"{tag}_{i}_points"] = model.feed_dict[i]["points"].numpy()            # (1,3,N) re-cropped, mean-subtracted
                out[f"{tag}_{i}_labels"] = model.feed_dict[i]["labels"].numpy().astype(np.int8)
                out[f"{tag}_{i}_nocs"] = model.npcs_feed_dict[i]["nocs"].numpy()
                seg = model.pred_dict["npcs_pred"][i]["seg"]
                top2 = torch.topk(seg, 2, dim=1)[0]
                out.setdefault(f"{tag}_min_softmax_gap", np.float32(np.inf))
                out[f"{tag}_min_softmax_gap"] = np.float32(min(float(out[f"{tag}_min_softmax_gap"]), float((top2[:, 0] - top2[:, 1]).min())))
        scales = np.stack([out[f"{tag}_{i}_scale"] for i in range(len(poses))])
        trans = np.stack([out[f"{tag}_{i}_translation"].reshape(3) for i in range(len(poses))])
        gt = np.stack([f["meta"]["nocs2camera"][0]["translation"].numpy().reshape(3) for f in data])
        print(f"{tag}: scales {scales.reshape(-1).round(4).tolist()}  |t - t_gt| {np.abs(trans - gt).max(1).round(4).tolist()}  "
              f"min softmax gap {float(out[f'{tag}_min_softmax_gap']):.2e}  object points per frame "
              f"{[int((out[f'{tag}_{i}_labels'] == 0).sum()) for i in range(1, len(poses))]}")
        assert scales.min() > 0.05 and scales.max() < 2.0, "trajectory left the physical regime"
        assert float(out[f"{tag}_min_softmax_gap"]) > 2e-5, "a point sits on a segmentation decision boundary: pick another seed"
    np.savez_compressed(HERE / "g15_otf_loop.npz", **out)
    print("wrote", HERE / "g15_otf_loop.npz")


if __name__ == "__main__":
    main()
