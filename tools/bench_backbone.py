"""BASELINE.json configs[4]: the 3-level set-abstraction backbone alone on synthetic 16384-point clouds.

SURVEY.md §8(d) C5: S-uni16k clouds (uniform in a cube), batch 64 over 8 GPUs = 8 clouds per GPU, sa1.npoint 2048 and
sa2.npoint 512 (radii / nsample / MLPs unchanged), plus the unscaled 512 / 128 sampling.  One "step" = one
`PointNet2Msg.forward` over the rank's clouds, replayed as a hipGraph; clouds never leave the GPU they live on (no
collective: N GPUs are N replicas of this loop, `--gpus N` under torch.distributed.run reports the aggregate).

    python tools/bench_backbone.py                       # both samplings, 8 clouds, one JSON line each
    python tools/bench_backbone.py --npoint 2048 512 --no-graph --steps 2     # what the PMC passes run

Per launch family the line carries the HIP-event time per step and, for the MFMA shared-MLP family, the flops of the
true channel counts (as bench.py does); for the drop-in geometry ops the §8(d) byte definitions at this shape.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from captra_amd import synthetic as clouds  # noqa: E402
from captra_amd.synthetic import make_state_dict  # noqa: E402

PEAK_MFMA_F32 = 157.3   # TFLOP/s, 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz
PEAK_HBM = 8000.0       # GB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--clouds", type=int, default=8, help="clouds per GPU (64 over 8 GPUs)")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--npoint", type=int, nargs=2, action="append", help="sa1.npoint sa2.npoint (repeatable)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipe", action="store_true", help="one graph per step (geometry, then MLPs) instead of captra_amd.graph.BackbonePipe "
                    "(batch t + 1's samplers / ball queries beside batch t's MLPs)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--xyz-feat", action="store_true", help="CoordNet's backbone (xyz as input features) instead of RotationNet's")
    args = ap.parse_args()
    samplings = args.npoint or [[2048, 512], [512, 128]]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)

    from captra_amd import _lib, fused
    from captra_amd.backbones import PointNet2Msg
    from captra_amd.configs import make_config

    B, N = args.clouds, args.points
    cloud = np.stack([clouds.s_uni(rank * B + i, N) for i in range(B)]).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(cloud.transpose(0, 2, 1))).to(device)

    for s1, s2 in samplings:
        cfg = copy.deepcopy(make_config("1"))
        cfg["pointnet"]["camera"]["sa1"]["npoint"] = s1
        cfg["pointnet"]["camera"]["sa2"]["npoint"] = s2
        net = PointNet2Msg(cfg, 128, use_xyz_feat=args.xyz_feat)
        net.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=21))
        net = net.to(device).eval()

        with torch.no_grad():
            for _ in range(3):
                out = net(x)
        torch.cuda.synchronize()
        graph = pipe = None
        if not args.no_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                net(x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                out = net(x)
            if not args.no_pipe:
                from captra_amd.graph import BackbonePipe
                pipe = BackbonePipe(net, x)
                graph.replay()
                ref = out.clone()
                for _ in range(3):
                    slot = pipe.push()
                assert torch.equal(pipe.output(slot), ref), "pipelined backbone differs from the one-graph step"

        def step():
            if pipe is not None:
                pipe.push()
            elif graph is not None:
                graph.replay()
            else:
                with torch.no_grad():
                    net(x)

        def timed(fn):
            for _ in range(args.warmup):
                fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            sync()
            return time.perf_counter() - t0

        def sync():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        elapsed = timed(step)
        one_graph_ms = 1e3 * timed(graph.replay) / args.steps if pipe is not None else None
        if pipe is not None:
            out = pipe.output((pipe.t - 1) % pipe.depth)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(out).all()

        fams = {}
        if not args.no_kernel_timing:
            _lib.prof_reset()
            _lib.prof_enable(True)
            fused.work_reset(True)
            with torch.no_grad():
                for _ in range(args.steps):
                    net(x)
            torch.cuda.synchronize()
            _lib.prof_enable(False)
            fused.WORK["on"] = False
            for name in _lib.prof_names():
                ms, n = _lib.prof_read(name)
                if n:
                    fams[name] = {"us_per_step": round(1e3 * ms / args.steps, 1), "launches_per_step": n // args.steps}
        if rank != 0:
            continue
        line = {
            "metric": "backbone clouds/sec (16384-pt clouds, 3-level set abstraction)" if N == 16384 else f"backbone clouds/sec ({N}-pt clouds)",
            "value": round(B * world * args.steps / elapsed, 2), "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S-uni16k, {N} pts/cloud, {B} clouds per GPU, sa1.npoint={s1}, sa2.npoint={s2}, "
                                   f"{'CoordNet' if args.xyz_feat else 'RotationNet'} backbone (BASELINE.json configs[4])",
                       "launch": ("two-stage pipeline of hipGraphs (BackbonePipe: geometry of batch t + 1 beside the MLPs of batch t)" if pipe is not None
                                  else "hipGraph replay" if graph is not None else "eager"),
                       "parallelism": f"{world} replicas, no collective"},
        }
        if one_graph_ms is not None:
            line["one_graph"] = {"ms_per_step": round(one_graph_ms, 3), "clouds/s": round(B * world / (one_graph_ms * 1e-3), 1),
                                 "note": "the same forward as ONE graph per batch (a batch's latency; samplers, then MLPs)"}
        if fams:
            mlp = [k for k in ("sa_scale_fused", "pointwise_mlp", "mlp_chain3", "coord_tail", "sa_group_mlp", "mlp_max") if k in fams]
            flops = sum(fused.WORK["flops"].get(k, 0.0) for k in mlp) / args.steps
            mlp_us = sum(fams[k]["us_per_step"] for k in mlp)
            line["kernels"] = fams
            if mlp_us:
                line["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_MFMA_F32,
                                    "achieved": round(flops / (mlp_us * 1e-6) / 1e12, 1),
                                    "frac": round(flops / (mlp_us * 1e-6) / 1e12 / PEAK_MFMA_F32, 3),
                                    "families": mlp, "us_per_step": round(mlp_us, 1),
                                    "share_of_kernel_time": round(mlp_us / sum(v["us_per_step"] for v in fams.values()), 3)}
                # HBM bytes per launch of the family from the committed counter passes of THIS workload (tools/profile_backbone.sh ->
                # profiles/<tag>_backbone16k_pmc.json), used only when taken on this tree's kernel sources (bench.pmc_traffic)
                import bench as _bench
                traffic, src, why = _bench.pmc_traffic(["sa_wave_kernel", "sa_wave_lds_kernel", "sa_wave_pipe_kernel", "sa_fused_kernel", "mlp_chain3_kernel",
                                                        "pw_direct_kernel", "pw_direct_max_kernel", "pw_mlp_kernel"], stem="backbone16k")
                line["roofline"]["traffic"] = None if traffic is None else round(traffic)
                line["roofline"]["traffic_source"] = f"profiles/{src}" if traffic is not None else f"null: {why}"
            # geometry ops of the fused path, §8(d) byte definitions at this shape (per step, all clouds)
            ks1, ks2 = (32, 64, 128), (64, 128)
            bq = B * (sum(12 * N + 12 * s1 + 4 * s1 * k for k in ks1) + sum(12 * s1 + 12 * s2 + 4 * s2 * k for k in ks2))
            if "ball_query" in fams:
                us = fams["ball_query"]["us_per_step"]
                line["ball_query"] = {"bytes_per_step": bq, "us_per_step": us, "GB/s": round(bq / us / 1e3, 1)}
            if "fps" in fams:
                us = fams["fps"]["us_per_step"]
                line["fps"] = {"us_per_step": us, "rounds": (s1 - 1) + (s2 - 1), "us_per_round": round(us / ((s1 - 1) + (s2 - 1)), 3)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
