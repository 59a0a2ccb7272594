"""CPU: the evaluation tables (captra_amd/pose_utils/bbox_utils.py, captra_amd/eval.py) against golden G10, produced by
the reference's own pose_utils/bbox_utils.py::eval_instance_part_iou and misc/eval/eval.py::get_joint_state
(tests/golden/make_golden_eval.py)."""
import pickle
from pathlib import Path

import numpy as np
import pytest
import torch

from captra_amd.pose_utils.bbox_utils import bbox_from_corners, eval_instance_part_iou, iou_3d, nocs_iou_3d
from tests.golden.make_golden_eval import make_inputs

G = np.load(Path(__file__).resolve().parent / "golden" / "g10_eval.npz")


@pytest.mark.parametrize("tag,P,sym,nocs", [("rigid_sym", 1, True, True), ("rigid", 1, False, True), ("arti", 4, False, False)])
def test_part_iou_vs_reference(tag, P, sym, nocs):
    gc, pc, gt, pred = make_inputs(11 + P + int(sym), P)
    got = eval_instance_part_iou(gc, pc, gt, pred, nocs=nocs, sym=sym)
    for name in ("npcs_iou", "iou", "gt_bbox_iou"):
        np.testing.assert_allclose(got[name], G[f"{tag}_{name}"], atol=1e-6, rtol=0, err_msg=name)


def test_iou_properties():
    box = bbox_from_corners(np.array([[-0.2, -0.1, -0.3], [0.2, 0.1, 0.3]], np.float32))
    assert abs(nocs_iou_3d(box, box) - 1.0) < 1e-6 and nocs_iou_3d(box, box + 10.0) == 0.0
    assert iou_3d(box, box) == 1.0 and iou_3d(box, box + 0.7) == 0.0      # disjoint, still resolved by the 50^3 grid
    assert iou_3d(box, box + 100.0) == 1.0   # the reference's grid artefact: no sample inside either box -> "both empty" -> 1
    shifted = iou_3d(box, box + np.array([0.2, 0.0, 0.0], np.float32))
    assert 0.0 < shifted < 1.0


def test_joint_state_and_eval_cli(tmp_path):
    """get_joint_state against the reference's values, then the whole `python -m captra_amd.eval` pass over a result pickle
    in the layout `captra_amd.track --save` writes."""
    from captra_amd import eval as ev
    gc, pc, gt, pred = make_inputs(11 + 4, 4)
    info = {"tree": [3, 3, 3, -1], "type": "prismatic", "main_axis": [2, 2, 2]}
    np.testing.assert_allclose(ev.get_joint_state(info, gt), G["arti_joint_state_gt"], atol=1e-6)
    np.testing.assert_allclose(ev.get_joint_state(info, pred), G["arti_joint_state_pred"], atol=1e-6)
    data_dir = tmp_path / "results" / "data"
    data_dir.mkdir(parents=True)
    frames = [gt, pred, pred]
    with open(data_dir / "inst0_track0.pkl", "wb") as f:
        pickle.dump({"pred": {"poses": frames, "corners": [None, pc, pc]}, "gt": {"poses": [gt, gt, gt], "corners": gc},
                     "frame_nums": [["0"], ["1"], ["2"]]}, f)
    avg = ev.main(["--obj_category", "drawers", "--obj_config", "obj_info_sapien.yml", "--experiment_dir", str(tmp_path)])
    np.testing.assert_allclose([avg[f"iou_{p}"] for p in range(4)], G["arti_iou"], atol=1e-6)
    np.testing.assert_allclose([avg[f"rdiff_{p}"] for p in range(4)], G["rot_diff_deg"], atol=1e-3)
    np.testing.assert_allclose([avg[f"theta_diff_{j}"] for j in range(3)],
                               np.abs(G["arti_joint_state_pred"] - G["arti_joint_state_gt"]), atol=1e-6)
    assert (tmp_path / "results" / "err.csv").exists() and (tmp_path / "results" / "err.pkl").exists()


def test_discarded_noise_draws_consume_the_generator_like_the_real_ones():
    """The track loop only CONSUMES the per-frame pose-noise draws (reference model.py:414-420 computes and drops them):
    consume_noise_draws must leave torch's generator exactly where add_noise_to_part_dof leaves it."""
    import torch
    from captra_amd.pose_utils.part_dof_utils import add_noise_to_part_dof, consume_noise_draws
    part = {"rotation": torch.eye(3).repeat(5, 2, 1, 1), "translation": torch.zeros(5, 2, 3, 1), "scale": torch.ones(5, 2)}
    for kind in ("normal", "uniform"):
        cfg = {"type": kind, "rotation": 0.1, "scale": 0.02, "translation": 0.03}
        torch.manual_seed(3)
        add_noise_to_part_dof(part, cfg)
        a = torch.rand(4)
        torch.manual_seed(3)
        consume_noise_draws(part, cfg)
        assert torch.equal(a, torch.rand(4)), kind


def test_config_dicts_agree_with_the_reference():
    """captra_amd.configs against the dicts the reference's get_config builds (tests/golden/make_golden_cfg.py) for the three
    experiment types and rigid / articulated categories: every key this package's configuration carries that the reference
    also has holds the same value (nested dicts: ours is a sub-dict of theirs — the reference tables also carry dataset
    bookkeeping and the hyper-parameters of modules outside the path)."""
    import json
    from pathlib import Path
    from captra_amd.configs import make_config
    ref_all = json.load(open(Path(__file__).resolve().parent / "golden" / "cfg_reference.json"))

    def sub(mine, ref, where):
        if isinstance(mine, dict) and isinstance(ref, dict):
            for k, v in mine.items():
                if k in ref:
                    sub(v, ref[k], f"{where}/{k}")
        else:
            assert mine == ref, (where, mine, ref)

    must_have = ("num_parts", "num_joints", "obj_tree", "obj_sym", "num_points", "data_radius", "network", "pose_perturb",
                 "batch_size", "pointnet", "obj_info", "obj_category")
    for key, ref in ref_all.items():
        config, cat, objcfg = key.split("|")
        mine = make_config(cat, objcfg, config=config)
        mine = json.loads(json.dumps({k: v for k, v in mine.items() if k not in ("device", "obj")}, default=str))
        assert all(k in mine and k in ref for k in must_have), key
        if config != "config_track.yml":
            assert all(k in mine for k in ("loss_weight", "pose_loss_type", "optimizer", "learning_rate", "weight_decay", "lr_policy",
                                           "lr_step_size", "lr_gamma", "lr_clip", "momentum_original", "momentum_decay",
                                           "momentum_step_size", "momentum_min", "weight_init", "total_epoch", "freq")), key
        sub({k: v for k, v in mine.items() if k not in ("experiment_dir", "num_expr", "root_dset")}, ref, key)
        for level in ("sa1", "sa2", "sa3", "fp3", "fp2", "fp1"):
            assert mine["pointnet"]["camera"][level] == ref["pointnet"]["camera"][level], (key, level)


def test_harness_flags_are_the_references():
    """track / train accept every flag of the reference's network/parse_args.py (names captured in the list below from
    parse_args.py:5-69) and turn `a/b` flags into cfg['a']['b'] overrides."""
    import argparse
    from captra_amd import parse_args as pa, track, train
    reference_flags = """config obj_config obj_category experiment_dir resume_epoch coord_exp/dir coord_exp/resume_epoch batch_size cuda_id
        total_epoch optimizer weight_decay learning_rate lr_policy lr_gamma lr_step_size lr_clip num_workers num_points data_radius
        dataset_length freq/save pointnet_cfg/camera network/type network/nocs_head_dims network/backbone_out_dim network/pwm_num save
        eval_train no_eval init_frame/gt loss_weight/rloss loss_weight/tloss loss_weight/sloss loss_weight/corner_loss
        loss_weight/nocs_loss loss_weight/nocs_dist_loss loss_weight/nocs_pwm_loss loss_weight/seg_loss pose_loss_type/r pose_loss_type/s
        pose_loss_type/t pose_loss_type/point pose_perturb/type pose_perturb/r pose_perturb/s pose_perturb/t nocs_otf
        track_cfg/gt_label track_cfg/nocs2d_label track_cfg/nocs2d_path""".split()
    for mod in (track, train):
        parser = mod.add_args(argparse.ArgumentParser())
        have = {a.dest for a in parser._actions}
        assert set(reference_flags) <= have, sorted(set(reference_flags) - have)
    from captra_amd.configs.config import get_config
    import tempfile
    ns = pa.add_args(argparse.ArgumentParser(), "config_rotnet.yml").parse_args(
        ["--obj_category", "3", "--experiment_dir", tempfile.mkdtemp(), "--loss_weight/rloss", "7.5", "--pose_loss_type/r", "l1",
         "--lr_step_size", "5", "--track_cfg/gt_label", "True", "--freq/save", "3"])
    cfg = get_config(ns, save=False)
    assert cfg["loss_weight"]["rloss"] == 7.5 and cfg["pose_loss_type"]["r"] == "l1" and cfg["lr_step_size"] == 5
    assert cfg["track_cfg"]["gt_label"] is True and cfg["freq"]["save"] == 3 and cfg["loss_weight"]["corner_loss"] == 1.0


def test_small_public_functions_vs_reference_g14():
    """The remaining public helpers of the reference's path modules against golden G14 (tests/golden/make_golden_api.py):
    the unmasked Procrustes family given a rotation (the free 3x3 solve runs the HIP kernel: GPU test), square_distance,
    compute_hard_miou_loss, get_pred_nocs_corners, get_posed_bbox_from_part, calc_part_iou_list."""
    import torch
    from pathlib import Path
    from captra_amd import loss as LS, pointnet_utils as PU
    from captra_amd.pose_utils import bbox_utils as BU, procrustes as P
    from tests.golden.make_golden_api import inputs
    g = np.load(Path(__file__).resolve().parent / "golden" / "g14_api.npz")
    d = inputs()
    t = torch.from_numpy
    src, tgt = t(d["src"]), t(d["tgt"])
    close = lambda a, k, tol=1e-5: np.testing.assert_allclose(np.asarray(a), g[k], atol=tol, rtol=1e-5, err_msg=k)
    close(P.scale_pts_batch(src, tgt), "scale_pts_batch")
    close(P.translate_pts_batch(src.transpose(-1, -2), tgt.transpose(-1, -2)), "translate_pts_batch")
    r2, t2 = P.transform_pts_2d_batch(src[..., [0, 2]], tgt[..., [0, 2]])
    close(r2, "t2d_rot"), close(t2, "t2d_trans")
    for tag, kw in (("given", {"rotation": t(d["given_rot"])}), ("given_sym", {"rotation": t(d["given_rot"]), "sym": True}),
                    ("given_scale", {"rotation": t(d["given_rot"]), "given_scale": torch.full((2, 3), 1.3)})):
        r, s, tr = P.transform_pts_batch(src, tgt, **kw)
        close(r, f"tpb_{tag}_rot"), close(s, f"tpb_{tag}_scale"), close(tr, f"tpb_{tag}_trans")
    close(PU.square_distance(src[:, 0], tgt[:, 1]), "square_distance", 1e-4)
    loss, miou = LS.compute_hard_miou_loss(t(d["labels_a"]), t(d["labels_b"]), 3, per_instance=True)
    close(loss, "hard_miou_loss"), close(miou, "hard_miou")
    close(BU.get_pred_nocs_corners(t(d["labels_a"]), t(d["nocs"]), 3), "pred_corners")
    pose = {k: t(v) for k, v in d["pose"].items()}
    pose2 = {k: t(v) for k, v in d["pose2"].items()}
    box1, box2 = BU.get_posed_bbox_from_part(pose, t(d["corners"])), BU.get_posed_bbox_from_part(pose2, t(d["corners"]))
    close(box1, "posed_bbox")
    mean, per = BU.calc_part_iou_list([box1], box2, separate="both", nocs=False)
    close([mean[p] for p in range(3)], "iou_mean", 2e-3), close(np.stack([per[p] for p in range(3)]), "iou_per", 2e-3)
    close([BU.calc_part_iou_list([box1], box2, separate=False, nocs=True)[p] for p in range(3)], "iou_mean_nocs")
    from captra_amd.pose_utils import rotations as RT
    rots = RT.unit_quaternion_to_matrix(RT.normalize(torch.from_numpy(np.random.default_rng(5).standard_normal((6, 4)).astype(np.float32))))
    close(RT.matrix_to_rotvec(rots), "rotvec"), close(RT.rotvec_to_matrix(RT.matrix_to_rotvec(rots)), "rotvec_back")
    for m in ("frob", "l1", "l2", "exp_l1", "exp_l2"):
        close(LS.rot_trace_loss(rots[:3], rots[3:], metric=m), f"rot_trace_{m}", 2e-5)
    close(LS.rot_yaxis_loss(rots[:3], rots[3:]), "rot_yaxis_l2")
