"""Pre-cropped trajectory files for the `track` harness.

The reference feeds `Trainer.test` a list over frames of frame dicts produced by its dataset classes
(network/data/dataset.py:71-91, 157-194; layout in SURVEY.md §8b "Loop API").  The datasets, the depth-crop
pipeline and cv2 are out of scope here (§8f row 1), so the harness reads the same frame dicts from one `.npz`
per trajectory -- the arrays a dataset item holds after cropping to N points:

    points       (T,3,N) f32   camera-frame points minus their mean
    points_mean  (T,3,1) f32
    labels       (T,N)   i64   part label per point (num_parts = background)
    nocs         (T,3,N) f32   ground-truth normalised coordinates
    rotation     (T,P,3,3) f32, translation (T,P,3,1) f32, scale (T,P) f32   ground-truth nocs2camera per part
    nocs_corners (P,2,3) f32
    paths        (T,) str      "…/<instance>/<track>/<frame>.<ext>" (drives the result file names, model.py:497-509)
    depth, mask  (T,H,W) uint16 mm / bool, optional: the frames' depth images and instance masks for `--nocs_otf True`
                               (the reference's meta['pre_fetched'], dataset.py:157-194)

`stack_trajectories` batches B trajectories of equal length into the (B, …) frame dicts `set_data` takes.
"""
from __future__ import annotations

import numpy as np
import torch

_KEYS = ("points", "points_mean", "labels", "nocs", "rotation", "translation", "scale", "nocs_corners", "paths")


def save_trajectory_npz(path: str, frames: list, b: int = 0) -> None:
    """Write trajectory `b` of a list of batched frame dicts (the structure above) to `path`."""
    P = len(frames[0]["meta"]["nocs2camera"])
    out = {
        "points": np.stack([f["points"][b].cpu().numpy() for f in frames]).astype(np.float32),
        "points_mean": np.stack([f["meta"]["points_mean"][b].cpu().numpy() for f in frames]).astype(np.float32),
        "labels": np.stack([f["labels"][b].cpu().numpy() for f in frames]).astype(np.int64),
        "nocs": np.stack([f["nocs"][b].cpu().numpy() for f in frames]).astype(np.float32),
        "rotation": np.stack([np.stack([f["meta"]["nocs2camera"][p]["rotation"][b].cpu().numpy() for p in range(P)]) for f in frames]),
        "translation": np.stack([np.stack([f["meta"]["nocs2camera"][p]["translation"][b].cpu().numpy() for p in range(P)]) for f in frames]),
        "scale": np.stack([np.stack([f["meta"]["nocs2camera"][p]["scale"][b].cpu().numpy() for p in range(P)]) for f in frames]),
        "nocs_corners": frames[0]["meta"]["nocs_corners"][b].cpu().numpy().astype(np.float32),
        "paths": np.array([f["meta"]["path"][b] for f in frames]),
    }
    if all("pre_fetched" in f["meta"] for f in frames):
        out["depth"] = np.stack([np.asarray(f["meta"]["pre_fetched"]["depth"][b]) for f in frames]).astype(np.uint16)
        out["mask"] = np.stack([np.asarray(f["meta"]["pre_fetched"]["mask"][b]) for f in frames]).astype(bool)
    np.savez(path, **out)


def load_trajectory_npz(path: str) -> dict:
    with np.load(path, allow_pickle=False) as z:
        missing = [k for k in _KEYS if k not in z.files]
        if missing:
            raise ValueError(f"{path}: not a trajectory file, missing {missing}")
        traj = {k: z[k] for k in _KEYS}
        for k in ("depth", "mask"):
            if k in z.files:
                traj[k] = z[k]
    T = traj["points"].shape[0]
    if not all(traj[k].shape[0] == T for k in ("points_mean", "labels", "nocs", "rotation", "translation", "scale", "paths")):
        raise ValueError(f"{path}: inconsistent frame counts")
    return traj


def stack_trajectories(trajs: list) -> list:
    """B trajectory dicts of equal length T -> list over T of batched frame dicts (CPU tensors)."""
    T = trajs[0]["points"].shape[0]
    if any(t["points"].shape != trajs[0]["points"].shape for t in trajs):
        raise ValueError("trajectories of one batch must share the frame count and the number of points")
    P = trajs[0]["rotation"].shape[1]
    frames = []
    for i in range(T):
        def cat(key, dtype=torch.float32):
            return torch.from_numpy(np.stack([t[key][i] for t in trajs])).to(dtype)
        poses = [{"rotation": cat("rotation")[:, p].contiguous(), "translation": cat("translation")[:, p].contiguous(),
                  "scale": cat("scale")[:, p].contiguous()} for p in range(P)]
        meta = {"path": [str(t["paths"][i]) for t in trajs], "nocs2camera": poses, "points_mean": cat("points_mean"),
                "nocs_corners": torch.from_numpy(np.stack([t["nocs_corners"] for t in trajs])).float()}
        if all("depth" in t and "mask" in t for t in trajs):
            meta["pre_fetched"] = {"depth": torch.from_numpy(np.stack([t["depth"][i].astype(np.int32) for t in trajs])),
                                   "mask": torch.from_numpy(np.stack([t["mask"][i] for t in trajs]))}
        frames.append({"points": cat("points"), "labels": cat("labels", torch.int64), "nocs": cat("nocs"), "meta": meta})
    return frames


def concat_frame_batches(batches: list) -> list:
    """[list over T of frame dicts with batch b_k] (equal T, N, P) -> one list over T with batch sum(b_k): the batch
    dimension of every tensor concatenated, path lists chained.  Used by the harness to build a batch out of
    independently generated / loaded trajectories, so that a trajectory's content does not depend on which rank or
    batch it lands in."""
    if len(batches) == 1:
        return batches[0]
    T = len(batches[0])
    if any(len(b) != T for b in batches):
        raise ValueError("trajectories of one batch must share the frame count")
    out = []
    for i in range(T):
        frames = [b[i] for b in batches]
        P = len(frames[0]["meta"]["nocs2camera"])
        meta = {"path": [p for f in frames for p in f["meta"]["path"]],
                "nocs2camera": [{k: torch.cat([f["meta"]["nocs2camera"][p][k] for f in frames]) for k in frames[0]["meta"]["nocs2camera"][p]}
                                for p in range(P)],
                "points_mean": torch.cat([f["meta"]["points_mean"] for f in frames]),
                "nocs_corners": torch.cat([f["meta"]["nocs_corners"] for f in frames])}
        if all("pre_fetched" in f["meta"] for f in frames):
            meta["pre_fetched"] = {k: torch.cat([torch.as_tensor(f["meta"]["pre_fetched"][k]) for f in frames]) for k in ("depth", "mask")}
        out.append({"points": torch.cat([f["points"] for f in frames]), "labels": torch.cat([f["labels"] for f in frames]),
                    "nocs": torch.cat([f["nocs"] for f in frames]), "meta": meta})
    return out
