"""Golden fixture G14: small public functions of the reference's path modules run on seeded inputs — the unmasked
Procrustes family (`scale_pts_batch`, `translate_pts_batch`, `transform_pts_2d_batch`, `transform_pts_batch`, with and
without a given rotation / symmetry), `square_distance`, `compute_hard_miou_loss`, `get_pred_nocs_corners`,
`calc_part_iou_list`, `get_posed_bbox_from_part`.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_api.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from tests.golden.make_golden import import_reference  # noqa: E402


def inputs():
    """Seeded inputs shared with the test."""
    rng = np.random.default_rng(77)
    src = rng.standard_normal((2, 3, 40, 3)).astype(np.float32)
    th = 0.7
    rot = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    tgt = (1.3 * src @ rot.T + np.array([0.1, -0.2, 0.3], np.float32) + 0.01 * rng.standard_normal(src.shape)).astype(np.float32)
    given_rot = np.broadcast_to(rot, (2, 3, 3, 3)).copy()
    labels_a = rng.integers(0, 3, (2, 500))
    labels_b = rng.integers(0, 3, (2, 500))
    nocs = (rng.random((2, 500, 3)) - 0.5).astype(np.float32)
    corners = np.stack([-np.abs(rng.random((2, 3, 3))) - 0.1, np.abs(rng.random((2, 3, 3))) + 0.1], axis=2).astype(np.float32)   # (B,P,2,3)
    pose = {"rotation": given_rot, "translation": rng.standard_normal((2, 3, 3, 1)).astype(np.float32) * 0.1,
            "scale": (1.0 + 0.1 * rng.random((2, 3))).astype(np.float32)}
    pose2 = {"rotation": given_rot, "translation": pose["translation"] + 0.02, "scale": pose["scale"] * 1.05}
    return dict(src=src, tgt=tgt, given_rot=given_rot, labels_a=labels_a, labels_b=labels_b, nocs=nocs, corners=corners, pose=pose, pose2=pose2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    pu = import_reference(args.ref)
    import procrustes as P
    import bbox_utils as BU
    import loss as LS
    d = inputs()
    t = torch.from_numpy
    out = {}
    src, tgt = t(d["src"]), t(d["tgt"])
    out["scale_pts_batch"] = P.scale_pts_batch(src, tgt).numpy()
    out["translate_pts_batch"] = P.translate_pts_batch(src.transpose(-1, -2), tgt.transpose(-1, -2)).numpy()
    r2, t2 = P.transform_pts_2d_batch(src[..., [0, 2]], tgt[..., [0, 2]])
    out["t2d_rot"], out["t2d_trans"] = r2.numpy(), t2.numpy()
    for tag, kw in (("free", {}), ("given", {"rotation": t(d["given_rot"])}), ("given_sym", {"rotation": t(d["given_rot"]), "sym": True}),
                    ("given_scale", {"rotation": t(d["given_rot"]), "given_scale": torch.full((2, 3), 1.3)})):
        r, s, tr = P.transform_pts_batch(src, tgt, **kw)
        out[f"tpb_{tag}_rot"], out[f"tpb_{tag}_scale"], out[f"tpb_{tag}_trans"] = r.numpy(), s.numpy(), tr.numpy()
    out["square_distance"] = pu.square_distance(src[:, 0], tgt[:, 1]).numpy()
    loss, miou = LS.compute_hard_miou_loss(t(d["labels_a"]), t(d["labels_b"]), 3, per_instance=True)
    out["hard_miou_loss"], out["hard_miou"] = loss.numpy(), miou.numpy()
    out["pred_corners"] = BU.get_pred_nocs_corners(t(d["labels_a"]), t(d["nocs"]), 3)
    pose = {k: t(v) for k, v in d["pose"].items()}
    pose2 = {k: t(v) for k, v in d["pose2"].items()}
    box1 = BU.get_posed_bbox_from_part(pose, t(d["corners"]))
    box2 = BU.get_posed_bbox_from_part(pose2, t(d["corners"]))
    out["posed_bbox"] = np.asarray(box1)
    mean, per = BU.calc_part_iou_list([box1], box2, separate="both", nocs=False)
    out["iou_mean"] = np.array([mean[p] for p in range(3)])
    out["iou_per"] = np.stack([per[p] for p in range(3)])
    mean_n = BU.calc_part_iou_list([box1], box2, separate=False, nocs=True)
    out["iou_mean_nocs"] = np.array([mean_n[p] for p in range(3)])
    import rotations as RT
    rots = RT.unit_quaternion_to_matrix(RT.normalize(torch.from_numpy(np.random.default_rng(5).standard_normal((6, 4)).astype(np.float32))))
    out["rotvec"] = RT.matrix_to_rotvec(rots).numpy()
    out["rotvec_back"] = RT.rotvec_to_matrix(RT.matrix_to_rotvec(rots)).numpy()
    for m in ("frob", "l1", "l2", "exp_l1", "exp_l2"):
        out[f"rot_trace_{m}"] = LS.rot_trace_loss(rots[:3], rots[3:], metric=m).numpy()
    out["rot_yaxis_l2"] = LS.rot_yaxis_loss(rots[:3], rots[3:]).numpy()
    np.savez_compressed(HERE / "g14_api.npz", **out)
    print("wrote", HERE / "g14_api.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
