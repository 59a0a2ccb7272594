"""Experiment: the B trajectories of a step as `lanes` independent sub-batches, each with its own captured hipGraph, replayed on
its own stream — one sub-batch's furthest-point sampling (one workgroup per cloud, B of 256 CUs busy) overlapping another's
MFMA kernels — against the single graph of bench.py.

    python tools/bench_ab.py [--batch 32] [--lanes 2] [--steps 40]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from bench import build_workload  # noqa: E402
from captra_amd.graph import TrackStepGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--overlap-nets", type=int, default=0)
    ap.add_argument("--join", type=int, default=0, help="1: the lanes start together and are joined on the main stream every step")
    ap.add_argument("--skew", type=int, default=0, help="delay lane l by l * skew 4096^2 matmuls (~1.5 ms each) once, before the warm-up")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg, sd, model, data = build_workload(args.batch, dev)
    B, L = args.batch, args.lanes
    nframes = len(model.feed_dict)
    res = {}
    for lanes, overlap in ((1, 1), (1, 0), (L, args.overlap_nets)):
        model.overlap_nets = bool(overlap)
        per = B // lanes
        sl = [slice(i * per, (i + 1) * per) for i in range(lanes)]
        pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
        f1 = model.feed_dict[1]
        graphs = [TrackStepGraph(model, f1["points"][s].contiguous(), f1["points_mean"][s].contiguous(), {k: v[s].contiguous() for k, v in pose.items()})
                  for s in sl]
        streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        poses = [{k: v[s].clone() for k, v in pose.items()} for s in sl]

        def step(i):
            f = 1 + i % (nframes - 1)
            fd = model.feed_dict[f]
            main = torch.cuda.current_stream(dev)
            for l in range(lanes):
                if args.join and lanes > 1:
                    streams[l].wait_stream(main)
                with torch.cuda.stream(streams[l]):
                    out = graphs[l].replay(fd["points"][sl[l]], fd["points_mean"][sl[l]], poses[l])
                    for k in out:
                        poses[l][k].copy_(out[k])
            if args.join and lanes > 1:
                for l in range(lanes):
                    main.wait_stream(streams[l])

        if lanes > 1 and args.skew:
            a = torch.randn(4096, 4096, device=dev)
            torch.cuda.synchronize()
            for l in range(1, lanes):
                with torch.cuda.stream(streams[l]):
                    for _ in range(l * args.skew):
                        a @ a
        for i in range(args.warmup):
            step(i)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0 - 0.0
        res[f"lanes{lanes}_overlap{overlap}"] = {"ms_per_step": round(1e3 * el / args.steps, 3), "frames_per_s": round(B * args.steps / el, 1)}
        del graphs
    print(json.dumps(res))


if __name__ == "__main__":
    main()
