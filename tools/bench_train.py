"""Training step throughput: `Trainer.update` of the CoordinateNet and RotationNet experiments on
synthetic batches, one process per GPU with the gradients averaged by one flat all-reduce per step.

    python tools/bench_train.py [--batch 12] [--steps 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tools/bench_train.py --gpus 8

One JSON line per experiment: samples/s (whole job), ms per step, and where a step's time goes (forward + losses,
backward, gradient exchange + Adam).  The training forward is NOT the fused inference path: it runs layer by layer under
autograd (torch convolutions / BatchNorm over the HIP sampling, grouping and interpolation operators).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from captra_amd.configs import make_config  # noqa: E402
from captra_amd.trainer import Trainer  # noqa: E402
from captra_amd import synthetic as clouds  # noqa: E402
from captra_amd.synthetic import make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=12, help="samples per GPU (the reference's batch_size)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the operators have no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)

    for name, config, cat, objcfg, kind in (("CoordinateNet (canon_coord), bottle", "config_coordnet.yml", "1", "obj_info_nocs.yml", "nocs"),
                                            ("RotationNet (rot), bottle", "config_rotnet.yml", "1", "obj_info_nocs.yml", "nocs"),
                                            ("RotationNet (rot), drawers (4 parts)", "config_rotnet.yml", "drawers", "obj_info_sapien.yml", "arti")):
        cfg = make_config(cat, objcfg, config=config)
        cfg["device"] = device
        trainer = Trainer(cfg)
        trainer.model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7))
        batches = [clouds.make_trajectory(kind, args.batch, 2, seed=10 * rank + i)[1] for i in range(2)]
        torch.manual_seed(rank)
        phases = {"forward+loss": 0.0, "backward": 0.0, "exchange+adam": 0.0}

        def step(i, timed):
            data = batches[i % len(batches)]
            t0 = time.perf_counter()
            trainer.optimizer.zero_grad()
            trainer.model.train()
            trainer.model.set_data(data)
            m = trainer.model
            if hasattr(m, "raw_feed_dict") and m.raw_feed_dict:
                m.prepare_data(m.raw_feed_dict)
                m.pred_dict = m.net(m.feed_dict, test_mode=False)
                m.compute_loss(test_mode=False)
            else:
                m.prepare_data()
                m.pred_dict = m.net(m.feed_dict)
                m.compute_loss()
            if timed:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            m.loss_dict["total_loss"].backward()
            if timed:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            if world > 1:
                from captra_amd.parallel import allreduce_gradients
                allreduce_gradients([p for p in m.parameters() if p.requires_grad], world)
            trainer.optimizer.step()
            if timed:
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                phases["forward+loss"] += t1 - t0
                phases["backward"] += t2 - t1
                phases["exchange+adam"] += t3 - t2

        def sync():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(args.warmup):
            step(i, False)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, False)
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        for i in range(3):
            step(i, True)
        loss = float(trainer.model.loss_dict["total_loss"].detach())
        assert np.isfinite(loss)
        if rank == 0:
            print(json.dumps({"metric": "training samples/sec (4096-pt clouds)", "value": round(args.batch * world * args.steps / elapsed, 2),
                              "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
                              "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{name}, batch {args.batch} per GPU, Adam, fp32", "parallelism": f"dp{world} (flat gradient all-reduce)"},
                              "ms_per_phase": {k: round(1e3 * v / 3, 2) for k, v in phases.items()}, "last_total_loss": round(loss, 4)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
