"""`python -m captra_amd.train`: the training loop (counterpart of the reference's network/train.py:30-98).

    python -m captra_amd.train --config config_coordnet.yml --obj_category 1 --experiment_dir runs/1_bottle_coord --data DIR
    python -m captra_amd.train --config config_rotnet.yml   --obj_category 1 --experiment_dir runs/1_bottle_rot   --data synthetic
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 -m captra_amd.train …

Same flow: `Trainer(cfg)`, `resume()`, per epoch `step_epoch()` then `update(data)` over the batches, averaged losses
logged as "Train <key> is <value>", checkpoint every `freq/save` epochs.  With several processes (one per GPU) every
rank trains on its own share of the samples and the gradients are averaged by one flat RCCL all-reduce per step.

Data: a training sample is one frame — `--data DIR` takes the frames of pre-cropped trajectory files
(`captra_amd/trajectory_io.py`), `--data synthetic[:nocs|:arti]` the seeded S-nocs / S-arti clouds.  The reference's
dataset classes (image decoding, on-disk layouts) are outside this build.
"""
from __future__ import annotations

import argparse
import glob
import os
from os.path import join as pjoin

import torch

from .configs.config import get_config
from .parse_args import add_args as _add_reference_args
from .trainer import Trainer
from .trajectory_io import load_trajectory_npz, stack_trajectories
from .utils import add_dict


def add_args(parser):
    """The reference's flags (captra_amd/parse_args.py) + the data source of this harness."""
    _add_reference_args(parser, default_config="config_coordnet.yml")
    parser.add_argument("--data", type=str, default="synthetic")
    parser.add_argument("--samples", type=int, default=48, help="synthetic data: samples per epoch and rank")
    return parser


def _batches(spec: str, batch: int, samples: int, rank: int):
    """-> list of frame dicts of `batch` samples each."""
    if spec.startswith("synthetic"):
        from captra_amd import synthetic as clouds
        kind = spec.split(":")[1] if ":" in spec else "nocs"
        return [clouds.make_trajectory(kind, batch, 2, seed=1000 * rank + i)[1] for i in range(max(samples // batch, 1))]
    files = sorted(glob.glob(pjoin(spec, "*.npz")))
    if not files:
        raise FileNotFoundError(f"no trajectory files under {spec}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    files = files[rank::world]
    out = []
    for i in range(0, len(files) - batch + 1, batch):
        frames = stack_trajectories([load_trajectory_npz(f) for f in files[i:i + batch]])
        out.extend(frames[1:])
    return out


def main():
    args = add_args(argparse.ArgumentParser()).parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    data_spec, samples = args.data, args.samples
    del args.data, args.samples
    if world > 1:
        args.cuda_id = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = get_config(args, save=rank == 0)
    if not torch.cuda.is_available():
        raise SystemExit("captra_amd.train needs a GPU: the operators have no CPU fallback")
    torch.cuda.set_device(cfg["device"])
    if world > 1:
        torch.distributed.init_process_group(backend="nccl", device_id=cfg["device"])
    trainer = Trainer(cfg)
    start = trainer.resume()
    batches = _batches(data_spec, cfg["batch_size"], samples, rank)
    for epoch in range(start, cfg["total_epoch"]):
        trainer.step_epoch()
        total = {}
        for data in batches:
            loss = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in trainer.update(data).items()}
            loss["cnt"] = 1
            add_dict(total, loss)
        cnt = total.pop("cnt")
        if rank == 0:
            for k, v in total.items():
                trainer.log_string("Train {} is {}".format(k, v / cnt))
            if (epoch + 1) % cfg["freq"]["save"] == 0:
                trainer.save()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
