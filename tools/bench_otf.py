"""On-the-fly re-crop of real NOCS tracking (`--nocs_otf True`): time of the crop + resample stage for a step of B
trajectories, one sampling launch per trajectory (the reference's structure) vs one ragged launch for the step, with
and without the spatially pruned sampler.

Usage: python tools/bench_otf.py [--batch 32]
"""
from __future__ import annotations

import argparse
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from captra_amd import _lib, nocs_otf  # noqa: E402
from captra_amd.synthetic import make_frame  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--radius", type=float, default=0.30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")

    items = []
    for b in range(args.batch):
        depth, mask, center, pose = make_frame(1 + b % 3)
        items.append((torch.from_numpy(depth.astype(np.int32)).to(dev), torch.from_numpy(mask).to(dev), center, args.radius, pose))
    np.random.seed(0)
    counts = [int(nocs_otf._candidate_cloud(*[x for j, x in enumerate(nocs_otf.crop_candidates(d, m, c, r, 4096)) if j != 1]).shape[0])
              for d, m, c, r, _ in items[:3]]
    print(f"candidate points per crop (first 3 frames): {counts}")

    def timed(fn, label):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        print(f"{label:98s} {ms:8.2f} ms / step of {args.batch} trajectories  ({ms / args.batch:6.3f} ms per frame)", flush=True)

    for pruned in (8192, 0):
        _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(pruned))
        tag = "pruned sampler" if pruned else "plain register-resident sampler"
        timed(lambda: [nocs_otf.full_data_from_depth(d, m, c, r, p, 4096) for d, m, c, r, p in items], f"one launch per trajectory, {tag}")
        timed(lambda: nocs_otf.full_data_batch(items, 4096, use_kernel=False), f"one ragged sampling launch per step, torch candidate extraction, {tag}")
        timed(lambda: nocs_otf.full_data_batch(items, 4096), f"crop kernel + one ragged sampling launch per step, {tag}")
    _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(8192))
    timed(lambda: [nocs_otf.crop_candidates(d, m, c, r, 4096) for d, m, c, r, _ in items], "candidate extraction alone (torch ops, host syncs)")
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(args.iters):
        nocs_otf.full_data_batch(items, 4096)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    for name in ("crop_ball", "fps"):
        ms, n = _lib.prof_read(name)
        print(f"  kernel {name:10s} {ms / max(n, 1):7.3f} ms per launch ({n} launches; HIP events on the launch stream)")


def track_loop(batch: int, frames: int = 12, configs=((False, False), (True, False), (True, True))):
    """The whole tracking loop with nocs_otf on (EvalTrackModel.forward: re-crop + hipGraph step per frame) vs off."""
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    from captra_amd import synthetic as clouds
    from captra_amd.synthetic import make_state_dict
    dev = torch.device("cuda:0")
    for otf, lanes, *rest in configs:
        overlap = rest[0] if rest else True
        cfg = make_config("1", experiment_dir="/tmp/captra_otf_bench", nocs_otf=otf, **{"init_frame/gt": True})
        cfg["device"] = dev
        trainer = Trainer(cfg)
        trainer.model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7))
        trainer.model.use_graph = True
        trainer.model.otf_lanes = bool(lanes)
        trainer.model.overlap_nets = bool(overlap)
        data = clouds.make_trajectory("nocs", batch, frames, seed=0)
        depth, mask, center, pose = make_frame(1)
        for f in data:
            f["meta"]["pre_fetched"] = {"depth": torch.from_numpy(np.stack([depth.astype(np.int32)] * batch)).to(dev),
                                        "mask": torch.from_numpy(np.stack([mask] * batch)).to(dev)}
            for p in f["meta"]["nocs2camera"]:
                p["rotation"] = torch.from_numpy(np.stack([pose["rotation"]] * batch)).float()
                p["translation"] = torch.from_numpy(np.stack([pose["translation"]] * batch)).float()
                p["scale"] = torch.full((batch,), float(pose["scale"]))
        np.random.seed(0)
        best = None
        for rep in range(3):
            trainer.model.eval()
            trainer.model.set_data(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.model.test(save=False, no_eval=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (frames - 1)
            best = dt if best is None or dt < best else best
        print(f"EvalTrackModel loop, nocs_otf={otf}{', two lanes half a frame apart' if lanes else ''}{'' if overlap else ', networks in sequence inside a lane'}: {best * 1e3:7.2f} ms per step of {batch} trajectories = {batch / best:7.0f} frames/s", flush=True)


def objects_loop(n_objects: int, batch: int = 32, frames: int = 8, loops: int = 3):
    """`n_objects` EvalTrackModel objects one after the other in ONE process, each timed on the two-lane
    re-crop loop (median of `loops` loops over 32 distinct depth frames): the schedule must not depend on how many streams /
    graphs the process created before."""
    from captra_amd.configs import make_config
    from captra_amd.synthetic import make_otf_trajectory, make_state_dict
    from captra_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    data = make_otf_trajectory(batch, frames, seed=1)
    for f in data:
        f["meta"]["pre_fetched"] = {k: v.to(dev) for k, v in f["meta"]["pre_fetched"].items()}
    out = []
    for k in range(n_objects):
        cfg = make_config("1", experiment_dir="/tmp/captra_otf_bench", nocs_otf=True, **{"init_frame/gt": True})
        cfg["device"] = dev
        trainer = Trainer(cfg)
        trainer.model.load_state_dict(make_state_dict({k2: tuple(v.shape) for k2, v in trainer.model.state_dict().items()}, seed=7))
        trainer.model.use_graph = True
        if "--no-overlap" in sys.argv:
            trainer.model.overlap_nets = False
        np.random.seed(0)
        ts = []
        for rep in range(loops + 1):
            trainer.model.eval()
            trainer.model.set_data(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.model.test(save=False, no_eval=True)
            torch.cuda.synchronize()
            if rep:                     # the first loop captures the graphs
                ts.append((time.perf_counter() - t0) / (frames - 1))
        ms = 1e3 * sorted(ts)[len(ts) // 2]
        out.append(ms)
        print(f"object {k:2d}: {ms:6.2f} ms per step (median of {loops}; all: {[round(1e3 * t, 2) for t in ts]})  schedule: {getattr(trainer.model, 'otf_schedule', '?')}", flush=True)
        del trainer
    print(f"{n_objects} objects: min {min(out):.2f}  max {max(out):.2f} ms per step")


if __name__ == "__main__":
    if "--objects" in sys.argv:
        objects_loop(int(sys.argv[sys.argv.index("--objects") + 1]))
        sys.exit(0)
    if "--ab-overlap" in sys.argv:      # same process, alternating: CoordNet || RotationNet inside a lane, or in sequence
        for rep in range(5):
            track_loop(32, configs=((True, True, True), (True, True, False)))
        track_loop(32, configs=((True, False, True), (True, False, False), (False, False, True), (False, False, False)))
        sys.exit(0)
    main()
    track_loop(32)
