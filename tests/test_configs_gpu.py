"""Acceptance gates of BASELINE.json's configs beyond the headline one.

configs[2] -- six rigid NOCS categories, bf16 MFMA operands in the shared MLPs -- is gated on 5 deg / 5 cm
(`eval_part_full`, reference pose_utils/part_dof_utils.py:54-67; categories from configs/obj_config/obj_info_nocs.yml:7-123):
the bf16 track loop must agree with the exact-fp32 loop, and with the reference's own loop where a golden exists (G9p),
within 5 deg and 5 cm on >= 99 % of the (frame, trajectory) pairs of free-running trajectories.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests import clouds
from tests.weights import make_physical_state_dict

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"

NOCS_CATEGORIES = {"1": "bottle", "2": "bowl", "3": "camera", "4": "can", "5": "laptop", "6": "mug"}


def _trainer(cat, device, mlp_dtype, wseed, objcfg="obj_info_nocs.yml", kind="nocs"):
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    cfg = make_config(cat, objcfg, experiment_dir="/tmp/captra_cfg2_test")
    cfg["mlp_dtype"] = mlp_dtype
    trainer = Trainer(cfg)
    shapes = {k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}
    trainer.model.load_state_dict(make_physical_state_dict(shapes, wseed, cfg["num_parts"], bool(cfg["obj_sym"]), kind))
    return trainer, cfg


def _agreement(cfg, poses_a, poses_b):
    """5deg5cm hits of pose_b against pose_a over frames 1.. (B*P per frame) -> (hit rate, max rdiff, max tdiff)."""
    from captra_amd.pose_utils.part_dof_utils import eval_part_model
    hits, rmax, tmax = [], 0.0, 0.0
    for a, b in zip(poses_a[1:], poses_b[1:]):
        d = eval_part_model(a, b, yaxis_only=bool(cfg["obj_sym"]))
        hits.append(torch.logical_and(d["rdiff"] <= 5.0, d["tdiff"] <= 0.05).float().reshape(-1))
        rmax, tmax = max(rmax, float(d["rdiff"].max())), max(tmax, float(d["tdiff"].max()))
    return float(torch.cat(hits).mean()), rmax, tmax


@pytest.mark.parametrize("cat", sorted(NOCS_CATEGORIES))
def test_configs2_bf16_track_loop_5deg5cm_gate_all_six_categories(device, cat):
    """Every rigid category (symmetric: 1, 2, 4; non-symmetric: 3, 5, 6), 8 trajectories x 8 frames free-running with
    cfg['mlp_dtype'] = 'bf16' against the exact-fp32 loop from the same seeded initial pose: >= 99 % of the (frame,
    trajectory) pairs within 5 deg / 5 cm; both loops stay in the physical regime."""
    data = clouds.make_trajectory("nocs", 8, 8, seed=30 + int(cat))
    poses = {}
    for dt in ("fp32", "bf16"):
        trainer, cfg = _trainer(cat, device, dt, wseed=40 + int(cat))
        assert trainer.model.mlp_dtype == dt
        torch.manual_seed(77)
        pred, loss = trainer.test(data, save=False, no_eval=False)
        poses[dt] = [{k: v.float().cpu() for k, v in p.items()} for p in pred["poses"]]
        assert all(torch.isfinite(v).all() for p in poses[dt] for v in p.values())
        assert min(float(p["scale"].min()) for p in poses[dt]) > 0.05
    for key in ("rotation", "translation", "scale"):      # same seed -> the same noisy initial pose
        np.testing.assert_array_equal(poses["fp32"][0][key].numpy(), poses["bf16"][0][key].numpy())
    rate, rmax, tmax = _agreement(cfg, poses["fp32"], poses["bf16"])
    assert rate >= 0.99, (NOCS_CATEGORIES[cat], rate, rmax, tmax)
    # and it is a bf16 run: the poses are not the fp32 run's bits
    assert any(not torch.equal(a["translation"], b["translation"]) for a, b in zip(poses["fp32"][1:], poses["bf16"][1:]))


@pytest.mark.parametrize("tag", ["bottle", "bowl", "camera", "can", "laptop", "mug", "drawers", "bottle5"])
def test_configs2_bf16_track_loop_vs_reference_golden_5deg5cm(device, tag):
    """Against the REFERENCE's own loop (goldens G9p: all six rigid categories of BASELINE.json configs[2], a five-trajectory batch,
    and the 4-part drawers of configs[3]): the bf16 loop's poses within 5 deg / 5 cm on every (frame, trajectory, part) pair."""
    cat, objcfg, kind, frames, batch, wseed, tseed = {**clouds.PHYSICAL_SETUPS, **clouds.PHYSICAL_SETUPS_MORE}[tag]
    trainer, cfg = _trainer(cat, device, "bf16", wseed, objcfg, kind)
    g = np.load(G / ("g9p_track.npz" if tag in clouds.PHYSICAL_SETUPS else "g9p_track_more.npz"))
    torch.manual_seed(tseed)
    pred, _ = trainer.test(clouds.make_trajectory(kind, batch, frames, seed=7), save=False, no_eval=True)
    ours = [{k: v.float().cpu() for k, v in p.items()} for p in pred["poses"]]
    ref = [{k: torch.from_numpy(g[f"{tag}_{i}_{k}"]) for k in ("rotation", "translation", "scale")} for i in range(frames)]
    rate, rmax, tmax = _agreement(cfg, ref, ours)
    assert rate == 1.0, (tag, rate, rmax, tmax)
