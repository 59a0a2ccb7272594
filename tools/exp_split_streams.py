"""Experiment: one hipGraph over B trajectories vs S graphs over B/S trajectories each, replayed on S streams
(independent trajectories: the latency-bound kernels of one sub-batch -- FPS, ball query, the 128-point layers --
can overlap the MFMA-bound kernels of another).  python tools/exp_split_streams.py [--batch 32] [--splits 1 2 4]"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from captra_amd.graph import TrackStepGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--splits", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, sd, model, data = bench.build_workload(a.batch, dev)
    f1 = model.feed_dict[1]
    pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
    for S in a.splits:
        bs = a.batch // S
        graphs, streams = [], []
        for s in range(S):
            sl = slice(s * bs, (s + 1) * bs)
            graphs.append(TrackStepGraph(model, f1["points"][sl].contiguous(), f1["points_mean"][sl].contiguous(),
                                         {k: v[sl].contiguous() for k, v in pose.items()}))
            streams.append(torch.cuda.Stream(device=dev))
        torch.cuda.synchronize()

        def step():
            cur = torch.cuda.current_stream(dev)
            for g, st in zip(graphs, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    g.graph.replay()
            for st in streams:
                cur.wait_stream(st)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"splits {S} x {bs} trajectories: {ms:.3f} ms/step  {a.batch / ms * 1e3:.0f} frames/s", flush=True)
        del graphs


if __name__ == "__main__":
    main()
