"""`python -m captra_amd.track`: the tracking harness (counterpart of the reference's network/test.py:33-101).

Builds the config from the same flags (`a/b` names override cfg['a']['b'], parse_args.py), constructs `Trainer`,
resumes the RotationNet experiment and the CoordNet experiment (`--coord_exp/dir`), walks the trajectories and
prints, per trajectory and overall, the reference's two throughput lines -- with a device synchronisation before
each clock read, which the reference lacks -- then the averaged pose errors; `--save` writes the per-trajectory
result pickles of model.py:482-509.

Data: `--data DIR` reads pre-cropped trajectories (`captra_amd/trajectory_io.py`: one .npz per trajectory);
`--data synthetic[:nocs|:arti]` generates the seeded S-nocs / S-arti trajectories of SURVEY.md §8d (needs this
repository's tests/ package).  `--nocs_otf True` re-crops every frame around the predicted pose on the device
(captra_amd/nocs_otf.py); the trajectory files then carry the frames' depth images and masks.  The reference's dataset
classes and image decoding (cv2) are outside this build.
"""
from __future__ import annotations

import argparse
import glob
import logging
import sys
import time
from os.path import join as pjoin

import torch

from .configs.config import get_config
from .parse_args import HARNESS_ONLY, add_args as _add_reference_args, boolean_string  # noqa: F401
from .trainer import Trainer
from .trajectory_io import load_trajectory_npz, stack_trajectories
from .utils import add_dict, ensure_dirs


def _is_number(v) -> bool:
    try:
        float(v)
        return True
    except (TypeError, ValueError):
        return False


def add_args(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    """The reference's flags (parse_args.py:5-69; captra_amd/parse_args.py), tracking configuration by default."""
    return _add_reference_args(parser, default_config="config_track.yml")


def parse_args(argv=None):
    parser = add_args(argparse.ArgumentParser(description=__doc__.split("\n")[0]))
    parser.add_argument("--data", type=str, default="synthetic", help="directory of trajectory .npz files, or synthetic[:nocs|:arti]")
    parser.add_argument("--num_traj", type=int, default=4, help="synthetic: number of trajectories")
    parser.add_argument("--num_frames", type=int, default=32, help="synthetic: frames per trajectory")
    parser.add_argument("--random_init", action="store_true", default=False,
                        help="no checkpoints: keep the freshly constructed (random) weights -- throughput runs only")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--mlp_dtype", default="fp32", choices=["fp32", "bf16"],
                        help="bf16: bf16 MFMA operands / fp32 accumulation in the shared MLPs (opt-in; default exact fp32)")
    parser.add_argument("--hipgraph", action="store_true", default=False,
                        help="replay one captured hipGraph per frame (same kernels, no per-launch host overhead)")
    return parser.parse_args(argv)


def iter_batches(args, cfg):
    """Yields lists over frames of batched frame dicts, cfg['batch_size'] trajectories at a time."""
    B = max(int(cfg.get("batch_size") or 1), 1)
    if args.data.startswith("synthetic"):
        kind = args.data.split(":")[1] if ":" in args.data else ("nocs" if cfg["num_parts"] == 1 else "arti")
        try:
            from tests import clouds
        except ImportError as e:  # pragma: no cover
            raise SystemExit("--data synthetic needs the repository's tests/ package on sys.path") from e
        done = 0
        while done < args.num_traj:
            b = min(B, args.num_traj - done)
            yield clouds.make_trajectory(kind, b, args.num_frames, seed=args.seed + done)
            done += b
        return
    files = sorted(glob.glob(pjoin(args.data, "*.npz")))
    if not files:
        raise SystemExit(f"no trajectory .npz files under {args.data}")
    trajs = [load_trajectory_npz(f) for f in files]
    trajs.sort(key=lambda t: t["points"].shape[0])          # equal-length trajectories batch together
    i = 0
    while i < len(trajs):
        j = i
        while j < len(trajs) and j - i < B and trajs[j]["points"].shape == trajs[i]["points"].shape:
            j += 1
        yield stack_trajectories(trajs[i:j])
        i = j


def main(argv=None) -> dict:
    args = parse_args(argv)
    data_args = {k: getattr(args, k) for k in ("data", "num_traj", "num_frames", "random_init", "seed", "hipgraph", "mlp_dtype")}
    for k in data_args:
        delattr(args, k)
    cfg = get_config(args, save=False)
    cfg["hipgraph"] = data_args["hipgraph"]
    from . import fused
    fused.MLP_DTYPE = data_args["mlp_dtype"]
    args = argparse.Namespace(**vars(args), **data_args)

    log_dir = pjoin(cfg["experiment_dir"], "log")
    ensure_dirs(log_dir)
    logger = logging.getLogger("TestModel")
    logger.setLevel(logging.INFO)
    handler = logging.FileHandler(pjoin(log_dir, "log_test.txt"))
    handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(handler)

    def log_string(msg):
        logger.info(msg)
        print(msg)

    log_string("PARAMETER ...")
    log_string({k: v for k, v in cfg.items() if k not in ("obj", "pointnet")})

    trainer = Trainer(cfg, logger)
    if args.random_init:
        trainer.model.to(cfg["device"])
        log_string("random-initialised weights (--random_init): pose errors are meaningless, throughput only")
    else:
        trainer.resume()

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    test_loss = {"cnt": 0}
    time_dict = {"data_proc": 0.0, "network": 0.0}
    total_frames = 0
    zero_time = time.time()
    for i, data in enumerate(iter_batches(args, cfg)):
        num_frames = len(data) * len(data[0]["points"])
        total_frames += num_frames
        print(f"Trajectory {i}, {num_frames:8} frames****************************")
        start_time = time.time()
        elapse = start_time - zero_time
        time_dict["data_proc"] += elapse
        print(f"Data Preprocessing: {elapse:8.2f}s {num_frames / max(elapse, 1e-9):8.2f}FPS")
        sync()
        start_time = time.time()
        pred_dict, loss_dict = trainer.test(data, save=cfg["save"], no_eval=cfg["no_eval"])
        sync()
        elapse = time.time() - start_time
        time_dict["network"] += elapse
        print(f"Network Forwarding: {elapse:8.2f}s {num_frames / max(elapse, 1e-9):8.2f}FPS")
        # per-trajectory averages (prediction and its initialisation); the per-frame tables stay in model.loss_dict
        flat = {f"{group}/{k}": float(v) for group in ("avg_pred", "avg_init")
                for k, v in (loss_dict.get(group) or {}).items() if _is_number(v)}
        flat["cnt"] = 1
        add_dict(test_loss, flat)
        zero_time = time.time()

    print(f"Overall, {total_frames:8} frames****************************")
    print(f"Data Preprocessing: {time_dict['data_proc']:8.2f}s {total_frames / max(time_dict['data_proc'], 1e-9):8.2f}FPS")
    print(f"Network Forwarding: {time_dict['network']:8.2f}s {total_frames / max(time_dict['network'], 1e-9):8.2f}FPS")
    cnt = max(test_loss.pop("cnt"), 1)
    summary = {}
    for key, val in test_loss.items():
        try:
            summary[key] = float(val) / cnt
        except (TypeError, ValueError):
            continue
        log_string("Test {} is {}".format(key, summary[key]))
    logger.removeHandler(handler)
    handler.close()
    return {"frames": total_frames, "network_s": time_dict["network"], "loss": summary}


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
