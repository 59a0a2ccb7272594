#!/usr/bin/env python
"""Headline benchmark: tracked frames/s on 4096-point clouds (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one rigid NOCS category (bottle: 1 part, symmetric),
4096 points per frame, 32 independent trajectories per GPU, fp32, synthetic S-nocs clouds
 resident in HBM, seeded random weights of the real architecture.
One STEP = one frame of the track loop for every trajectory of the rank: CoordNet (canonicalise,
PointNet++ MSG backbone, seg/NOCS heads) -> labels -> RotationNet (P backbone clouds, rotation heads,
masked pooling, orthogonalisation) -> per-part Procrustes scale/translation fit; the pose feeds
the next step (frame i needs pose i-1, reference model.py:408-478).  Trajectories shard over
GPUs with no data-path dependency; the per-frame pose records are all-gathered with RCCL so that
every rank holds the full result ("weak" scaling: 32 trajectories per GPU).

CoordNet and RotationNet run side by side on two streams (two branches of the replayed hipGraph; `--no-overlap` puts them
back one after the other), and from 32 trajectories per GPU on the rank's trajectories run as two free-running lanes of half the
batch each (`--lanes`, captra_amd.graph.TrackLanes: every lane replays its own captured step on its own stream and hands its pose
over to itself; the frame's poses reach the exchange through a ring of records).  A step still passes every trajectory of the rank
through one frame, and the timed region ends with a device-wide synchronize + barrier.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` describes the dominant kernel
family of the step, measured with HIP events on the launch stream over the same steps launched eagerly right after the
timed region, networks one after the other (a launch's duration is then the kernel's own);
`cpu_baseline` is the CPU oracle (oracle/, a port of the reference's CPU path) timed on the host
cores of rank 0 at N=1 on a bounded sample of the same workload at batch 1.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak (only for --mlp-dtype bf16 runs)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


WORKLOADS = {  # name: (obj_category, obj_config, synthetic trajectory kind, description)
    "bottle": ("1", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'bottle' (1 part, symmetric)"),
    "bowl": ("2", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'bowl' (1 part, symmetric)"),
    "camera": ("3", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'camera' (1 part, non-symmetric)"),
    "can": ("4", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'can' (1 part, symmetric)"),
    "laptop": ("5", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'laptop' (1 part, non-symmetric)"),
    "mug": ("6", "obj_info_nocs.yml", "nocs", "NOCS-REAL275-shaped rigid category 'mug' (1 part, non-symmetric)"),
    "drawers": ("drawers", "obj_info_sapien.yml", "arti", "SAPIEN-shaped articulated category 'drawers' (4 parts)"),
}


MIX6 = ["bottle", "bowl", "camera", "can", "laptop", "mug"]


def csrc_fingerprint() -> str:
    """sha1 (16 hex) over the kernel sources (captra_amd/csrc/*.hip, *.h, *.cpp) and the build recipe: the identity of the
    code a committed counter file was profiled on (there is no .git on the GPU box to ask)."""
    import hashlib
    h = hashlib.sha1()
    csrc = ROOT / "captra_amd" / "csrc"
    for p in sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.h")) + list(csrc.glob("*.cpp")) + [ROOT / "captra_amd" / "build.py"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def build_workload(batch: int, device, frames: int = 8, category: str = "bottle", traj_seed: int = 0, mlp_dtype: str = "fp32"):
    """`batch` DISTINCT synthetic trajectories (clouds seeded per trajectory; `traj_seed` = the rank, so that no two ranks
    track the same objects) with physical-regime seeded weights (tests/weights.make_physical_state_dict: the random SA / FP
    stack of make_state_dict + a planted coordinate pass-through, so the loop tracks -- positive scales, small rotations --
    and `pose_match` can be asked of the timed trajectories themselves)."""
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    from captra_amd import synthetic as clouds
    from captra_amd.synthetic import make_physical_state_dict

    obj_category, obj_config, kind, _ = WORKLOADS[category]
    cfg = make_config(obj_category, obj_config, experiment_dir="/tmp/captra_bench")
    cfg["device"] = device
    cfg["mlp_dtype"] = mlp_dtype
    trainer = Trainer(cfg)
    shapes = {k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}
    sd = make_physical_state_dict(shapes, 7, cfg["num_parts"], bool(cfg["obj_sym"]), kind)
    trainer.model.load_state_dict(sd)
    model = trainer.model.eval()
    data = clouds.make_trajectory(kind, batch, frames, seed=10 + traj_seed)     # cloud b of this rank: seed (10 + rank) * 100 + b
    model.set_data(data)           # host -> HBM once, before any timed region
    return cfg, sd, model, data


def _set_cpu_threads(n: int) -> None:
    torch.set_num_threads(n)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(ctypes.c_int(n))   # the C oracle's OpenMP loops
    except OSError:
        pass


def cpu_baseline(cfg, sd, budget_s: float = 15.0):
    """The CPU oracle (port of the reference's CPU path: C geometry + torch-CPU shared MLPs) on
    batch 1 of the same workload, timed on this host's cores.  The thread count that runs a frame
    fastest (of 8/16/32/64, capped by the host) is used and reported as `cores`."""
    from oracle import model as OM
    from captra_amd import synthetic as clouds
    data = clouds.make_trajectory("nocs", 1, 6, seed=0)
    pose0 = {k: v.numpy() for k, v in
             {"rotation": data[0]["meta"]["nocs2camera"][0]["rotation"].unsqueeze(1),
              "translation": data[0]["meta"]["nocs2camera"][0]["translation"].unsqueeze(1),
              "scale": data[0]["meta"]["nocs2camera"][0]["scale"].unsqueeze(1)}.items()}

    def one(i, pose):
        f = data[1 + i % 5]
        return OM.track_step(sd, cfg, f["points"].numpy(), f["meta"]["points_mean"].numpy(), pose, "torch")[0]

    ncpu = os.cpu_count() or 8
    best, best_t = None, 1e30
    for n in [c for c in (8, 16, 32, 64) if c <= ncpu] or [ncpu]:
        _set_cpu_threads(n)
        one(0, pose0)                       # warm-up at this thread count
        t0 = time.time()
        one(1, pose0)
        dt = time.time() - t0
        if dt < best_t:
            best, best_t = n, dt
    _set_cpu_threads(best)
    pose, frames, t0 = pose0, 0, time.time()
    while True:
        pose = one(frames, pose)
        frames += 1
        if time.time() - t0 > budget_s or frames >= 200:
            break
    dt = time.time() - t0
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": int(best), "kind": "port",
            "sample": f"{frames} frames of the bottle workload at batch 1 (4096 pts), oracle/model.py track_step, "
                      f"{dt:.1f} s wall with {best} threads (fastest of 8/16/32/64), host has {ncpu} logical cores"}


def otf_leg(batch: int, device, frames: int = 10, reps: int = 5):
    """The loop every `scripts/track/nocs/*.sh` of the reference actually runs: `EvalTrackModel.test` with `nocs_otf=True` — per
    frame the on-device re-crop of the depth image around the previous pose (csrc/crop.hip), the 15 k -> 4096 furthest-point
    sampling (the pruned ragged sampler) and the captured step — on `batch` DISTINCT synthetic depth frames whose object
    drifts from frame to frame (captra_amd.synthetic.make_otf_trajectory, the generator of golden G15), Python included.
    Reported beside the headline because it is about 70 % of it; `two_lanes` (the default from 32 trajectories on,
    cfg['otf_lanes']) runs the batch as two sub-batches half a frame apart — one samples while the other runs its networks —
    with bit-identical poses; `single_batch` is the same loop with that switched off.  Median of `reps` loops each."""
    import tempfile
    from captra_amd.configs import make_config
    from captra_amd.synthetic import make_otf_trajectory, make_state_dict
    from captra_amd.trainer import Trainer
    data = make_otf_trajectory(batch, frames, seed=1)
    for f in data:
        f["meta"]["pre_fetched"] = {k: v.to(device) for k, v in f["meta"]["pre_fetched"].items()}
    out = {"workload": f"EvalTrackModel.test, nocs_otf=True, bottle, {batch} trajectories x {frames} frames, ~15 k candidates per crop -> 4096", "unit": "frames/s"}
    for key, lanes in (("two_lanes", True), ("single_batch", False)):
        cfg = make_config("1", experiment_dir=tempfile.mkdtemp(prefix="captra_bench_otf_"), nocs_otf=True, **{"init_frame/gt": True})
        cfg["device"] = device
        trainer = Trainer(cfg)
        trainer.model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7))
        trainer.model.use_graph = True
        trainer.model.otf_lanes = lanes
        np.random.seed(0)
        times = []
        for rep in range(reps + 1):
            trainer.model.eval()
            trainer.model.set_data(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.model.test(save=False, no_eval=True)
            torch.cuda.synchronize()
            if rep:                                  # the first loop captures the step
                times.append((time.perf_counter() - t0) / (frames - 1))
        times.sort()
        med = times[(len(times) - 1) // 2]
        out[key] = {"value": round(batch / med, 1), "ms_per_step": round(med * 1e3, 3), "ms_per_step_min": round(times[0] * 1e3, 3),
                    "ms_per_step_max": round(times[-1] * 1e3, 3)}
        del trainer
    out["value"] = out["two_lanes"]["value"]
    out["note"] = f"value = two lanes (default); median of {reps} loops; not the headline (configs[1] feeds pre-cropped clouds)"
    return out


def b1_leg(device, steps: int = 200, otf_frames: int = 24, reps: int = 5):
    """Single-trajectory latency, the reference's own measuring convention (`--batch_size=1`, /root/reference README.md:267; the
    on-the-fly crop asserts batch 1, model.py:319; BASELINE.json configs[0]'s workload): ms per frame at B = 1 for (a) the
    pre-cropped step (captured graph, pose chained frame to frame) with its per-family kernel times from an eager pass, and
    (b) the `nocs_otf=True` loop (re-crop + 15 k -> 4096 sampling + step per frame, Python included)."""
    import tempfile
    from captra_amd import _lib, fused
    from captra_amd.configs import make_config
    from captra_amd.graph import TrackStepGraph
    from captra_amd.synthetic import make_otf_trajectory, make_state_dict
    from captra_amd.trainer import Trainer
    cfg, sd, model, data = build_workload(1, device, frames=8)
    pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
    f1 = model.feed_dict[1]
    graph = TrackStepGraph(model, f1["points"], f1["points_mean"], pose)
    nfr = len(model.feed_dict)
    times = []
    for rep in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fd = model.feed_dict[1 + i % (nfr - 1)]
            pose = graph.replay(fd["points"], fd["points_mean"], pose)
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) / steps)
    times.sort()
    pre = times[(len(times) - 1) // 2]
    # kernel families of one frame (eager launches, networks one after the other: a launch's duration is the kernel's own)
    model.overlap_nets = False
    p2 = {k: v.clone() for k, v in pose.items()}
    _lib.prof_reset()
    _lib.prof_enable(True)
    with torch.no_grad():
        for i in range(20):
            _, p2 = model.track_step(model.feed_dict[1 + i % (nfr - 1)], model.npcs_feed_dict[1 + i % (nfr - 1)], p2)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    fams = {n: _lib.prof_read(n)[0] / 20 for n in _lib.prof_names() if _lib.prof_read(n)[1]}
    out = {"unit": "ms per frame", "convention": "reference README.md:267 (--batch_size=1), model.py:319",
           "pre_cropped": {"ms_per_frame": round(pre * 1e3, 4), "frames_per_s": round(1.0 / pre, 1), "launch": "hipGraph replay, pose chained",
                           "kernel_ms_per_frame_networks_in_sequence": {k: round(v, 3) for k, v in sorted(fams.items(), key=lambda kv: -kv[1])[:8]}}}
    del graph, model
    cfg = make_config("1", experiment_dir=tempfile.mkdtemp(prefix="captra_bench_b1_"), nocs_otf=True, **{"init_frame/gt": True})
    cfg["device"] = device
    trainer = Trainer(cfg)
    trainer.model.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7))
    trainer.model.use_graph = True
    otf = make_otf_trajectory(1, otf_frames, seed=1, step_px=2.0)
    for f in otf:
        f["meta"]["pre_fetched"] = {k: v.to(device) for k, v in f["meta"]["pre_fetched"].items()}
    np.random.seed(0)
    times = []
    for rep in range(reps + 1):
        trainer.model.eval()
        trainer.model.set_data(otf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.model.test(save=False, no_eval=True)
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) / (otf_frames - 1))
    times.sort()
    med = times[(len(times) - 1) // 2]
    _lib.prof_reset()
    _lib.prof_enable(True)
    trainer.model.set_data(otf)
    trainer.model.test(save=False, no_eval=True)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    crop = {n: _lib.prof_read(n)[0] / (otf_frames - 1) for n in ("crop_ball", "fps")}
    out["nocs_otf"] = {"ms_per_frame": round(med * 1e3, 4), "frames_per_s": round(1.0 / med, 1),
                       "launch": "EvalTrackModel.test, nocs_otf=True, Python included",
                       "kernel_ms_per_frame": {"crop_ball": round(crop["crop_ball"], 4),
                                               "fps": round(crop["fps"], 4)}}
    out["note"] = f"median of {reps} runs each"
    return out


def ball_query_scanned_fraction(points_cn, pn_cfg) -> float:
    """Share of the N x M pair tests per (cloud, level) the index-order scan actually makes before its early exit (every radius'
    list full: bq_scan.h), on THESE clouds -- a torch restatement of the exit rule on the sampled centres: a centre's scan ends
    with the 64-point chunk in which its slowest list takes its K-th hit (at N when a list never fills).  Measurement only."""
    import torch
    from captra_amd import fused
    xyz = points_cn.transpose(1, 2).contiguous()                       # (B,N,3)
    scanned = total = 0.0
    for lvl in ("sa1", "sa2"):
        c = pn_cfg[lvl]
        _, new_n3, _ = fused.fps_gather(xyz, int(c["npoint"]))
        d2 = ((new_n3[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)        # (B,M,N)
        n = xyz.shape[1]
        end = torch.zeros(d2.shape[:2], dtype=torch.long, device=xyz.device)
        for r, k in zip(c["radius_list"], c["nsample_list"]):
            cnt = (d2 < float(r) ** 2).cumsum(-1, dtype=torch.int32)
            full = cnt[..., -1] >= int(k)
            pos = (cnt >= int(k)).to(torch.int8).argmax(-1) + 1           # points read up to and including the K-th hit
            end = torch.maximum(end, torch.where(full, pos, torch.full_like(pos, n)))
        end = torch.clamp((end + 63) // 64 * 64, max=n)
        scanned += float(end.sum().item())
        total += float(end.numel() * n)
        xyz = new_n3.contiguous()
    return scanned / max(total, 1.0)


def hbm_ops_roofline(batch: int, device, reps: int = 5):
    """The drop-in ops ball_query + group_points (SURVEY.md §8d "materialised-op" byte definition: ball query
    12N + 12M + 4MK, group 4CN + 4MK + 4CMK bytes per cloud) on the workload's SA1 / SA2 shapes for one frame
    of both nets, timed with HIP events on the launch stream (captra_prof).  The product path does not run
    group_points at all (the SA kernels gather inside their operand load): this is the op-level figure."""
    import torch
    from captra_amd import _lib
    from captra_amd import pointnet2_cuda as pc
    from captra_amd import synthetic as clouds
    B = batch
    pts = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)])).to(device).contiguous()   # (B,N,3), B distinct clouds
    gen = torch.Generator(device="cpu").manual_seed(3)
    levels = [  # (N, M, [(radius, K)], feature channel counts grouped per frame: rot net xyz, coord net xyz + C0)
        (4096, 512, [(0.05, 32), (0.1, 64), (0.2, 128)], [3, 3, 3]),
        (512, 128, [(0.2, 64), (0.4, 128)], [3, 320, 3, 320]),
    ]
    nbytes = {"ball_query": 0.0, "group_points": 0.0}
    work = []
    xyz = pts
    for (n, m, rk, chans) in levels:
        xyz = xyz[:, :n].contiguous()
        new_xyz = xyz[:, :m].contiguous()
        feats = {c: torch.randn(B, c, n, generator=gen).to(device) for c in set(chans)}
        for (r, k) in rk:
            idx = torch.zeros(B, m, k, dtype=torch.int32, device=device)
            outs = [torch.empty(B, c, m, k, device=device) for c in chans]      # one output tensor per job (no two jobs share lines)
            work.append((n, m, r, k, new_xyz, xyz, idx, feats, outs, chans))
            nbytes["ball_query"] += B * (12.0 * n + 12.0 * m + 4.0 * m * k)
            for c in chans:
                nbytes["group_points"] += B * (4.0 * c * n + 4.0 * m * k + 4.0 * c * m * k)

    from captra_amd import fused as _fused

    def run_multi_group():
        """one ball-query launch and ONE group launch per level (captra_ball_query_multi + captra_group_points_multi)"""
        for lvl_n in sorted({w[0] for w in work}, reverse=True):
            lvl = [w for w in work if w[0] == lvl_n]
            radii = (ctypes.c_float * len(lvl))(*[w[2] for w in lvl])
            ks = (ctypes.c_int * len(lvl))(*[w[3] for w in lvl])
            ptrs = (ctypes.c_void_p * len(lvl))(*[w[6].data_ptr() for w in lvl])
            _lib.call("captra_ball_query_multi", B, lvl_n, lvl[0][1], len(lvl), ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(ks, ctypes.c_void_p),
                      lvl[0][4].data_ptr(), lvl[0][5].data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p))
            # a network's own output tensor per (radius, feature tensor): the coord net groups xyz twice (coordinates and features)
            pts, idxs, outs_l = [], [], []
            for w in lvl:
                for ci, c in enumerate(w[9]):
                    pts.append(w[7][c]); idxs.append(w[6]); outs_l.append(w[8][ci])
            _fused.group_points_multi(pts, idxs, outs_l)

    def run(multi=False):
        done = set()
        for (n, m, r, k, new_xyz, xyz_l, idx, feats, outs, chans) in work:
            if not multi:
                pc.ball_query_wrapper(B, n, m, r, k, new_xyz, xyz_l, idx)
            elif n not in done:      # one scan per level for all its radii (captra_ball_query_multi, what the host mirror calls)
                done.add(n)
                lvl = [w for w in work if w[0] == n]
                radii = (ctypes.c_float * len(lvl))(*[w[2] for w in lvl])
                ks = (ctypes.c_int * len(lvl))(*[w[3] for w in lvl])
                ptrs = (ctypes.c_void_p * len(lvl))(*[w[6].data_ptr() for w in lvl])
                _lib.call("captra_ball_query_multi", B, n, m, len(lvl), ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(ks, ctypes.c_void_p),
                          new_xyz.data_ptr(), xyz_l.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p))
            for ci, c in enumerate(chans):
                pc.group_points_wrapper(B, c, n, m, k, feats[c], idx, outs[ci])

    run()
    torch.cuda.synchronize()
    ref_idx = [w[6].clone() for w in work]
    for w in work:
        w[6].zero_()
    run(multi=True)
    torch.cuda.synchronize()
    assert all(torch.equal(a, w[6]) for a, w in zip(ref_idx, work)), "multi-radius ball query differs from the single-radius op"
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        run(multi=True)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms_multi = {k: _lib.prof_read(k)[0] / reps for k in nbytes}
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms = {k: _lib.prof_read(k)[0] / reps for k in nbytes}
    run_multi_group()
    torch.cuda.synchronize()
    assert all(torch.equal(a, w[6]) for a, w in zip(ref_idx, work))
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        run_multi_group()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms_lvl = {k: _lib.prof_read(k)[0] / reps for k in nbytes}

    def sequence_ms(fn):
        """the whole job between ONE pair of events on the launch stream: what the sequence takes end to end (the per-op figures
        above bracket every launch on its own, which puts an event's signal + dispatch between any two kernels)"""
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        best = []
        for _ in range(3):
            s0.record()
            for _ in range(reps):
                fn()
            s1.record()
            torch.cuda.synchronize()
            best.append(s0.elapsed_time(s1) / reps)
        return sorted(best)[1]
    seq_plain, seq_multi, seq_lvl = sequence_ms(run), sequence_ms(lambda: run(multi=True)), sequence_ms(run_multi_group)

    # the reference's QueryAndGroup MODULE (pointnet2_utils.py:274-310), one call per (network, radius) as PointNet2Msg's levels make
    # them: RotationNet groups the coordinates (SA1) / 320 features + coordinates (SA2), CoordinateNet its 3 features + coordinates
    # (SA1) / 320 + coordinates (SA2) -- each call ONE launch here (captra_query_and_group: search, both gathers, centre subtraction
    # and concat; the lists never leave LDS).  Bytes: every call's own ball query + its grouping jobs, materialised-op definition.
    from captra_amd.pointnet_lib import pointnet2_utils as pn
    qg_calls, qg_bytes = [], 0.0
    for (n, m, r, k, new_xyz, xyz_l, idx, feats, outs, chans) in work:
        for feat_c in ((None, 3) if n == 4096 else (320, 320)):
            qg_calls.append((pn.QueryAndGroup(r, k, use_xyz=True), xyz_l, new_xyz, None if feat_c is None else feats[feat_c]))
            qg_bytes += B * (12.0 * n + 12.0 * m + 4.0 * m * k)
            for c in ((3,) if feat_c is None else (3, feat_c)):
                qg_bytes += B * (4.0 * c * n + 4.0 * m * k + 4.0 * c * m * k)

    def run_qg():
        for mod, x, nx, f in qg_calls:
            mod(x, nx, f)

    # one call against the two ops + torch glue it replaces (values, bit for bit), then the timing
    mod, x, nx, f = qg_calls[-1]
    chk_idx = pn.ball_query(mod.radius, mod.nsample, x, nx)
    chk = torch.cat([pn.grouping_operation(f, chk_idx), pn.grouping_operation(x.transpose(1, 2).contiguous(), chk_idx) - nx.transpose(1, 2).unsqueeze(-1)], dim=1)
    assert torch.equal(mod(x, nx, f), chk), "one-launch QueryAndGroup differs from ball_query + grouping_operation"
    del chk, chk_idx
    seq_qg = sequence_ms(run_qg)
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        run_qg()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms_qg = _lib.prof_read("query_and_group")[0] / reps
    # what a plain write stream reaches on this box, measured in this run: the ceiling the group op's 4*C*M*K output bytes face
    probe = torch.empty(128 << 20, dtype=torch.float32, device=device)
    probe.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        probe.fill_(2.0)
    e1.record()
    torch.cuda.synchronize()
    fill_gbs = 5 * probe.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del probe
    tot_b, tot_ms = sum(nbytes.values()), sum(ms.values())
    frac = lambda nb, t_ms: round(nb / (t_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    # (what each figure is: README "bench line" / DESIGN.md section 5 -- kept out of the line, which has to fit the driver's 8 KB tail)
    out = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
           "achieved": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "frac": frac(tot_b, tot_ms),
           "bytes_per_frame": round(tot_b / B), "us_per_frame": round(1e3 * tot_ms / B, 2),
           "ops": {k: {"GB/s": round(nbytes[k] / (ms[k] * 1e-3) / 1e9, 1), "ms": round(ms[k], 3)} for k in nbytes},
           "sequence": {"ms": round(seq_plain, 3), "frac": frac(tot_b, seq_plain),
                        "multi_radius_ms": round(seq_multi, 3), "multi_radius_frac": frac(tot_b, seq_multi),
                        "per_level_ms": round(seq_lvl, 3), "per_level_frac": frac(tot_b, seq_lvl)},
           "per_level": {"frac": frac(tot_b, sum(ms_lvl.values())), "GB/s": round(tot_b / (sum(ms_lvl.values()) * 1e-3) / 1e9, 1),
                         "ball_query_ms": round(ms_lvl["ball_query"], 3), "group_points_ms": round(ms_lvl["group_points"], 3)},
           "query_and_group": {"launches": len(qg_calls), "bytes_per_frame": round(qg_bytes / B), "ms": round(ms_qg, 3), "frac": frac(qg_bytes, ms_qg),
                               "GB/s": round(qg_bytes / (ms_qg * 1e-3) / 1e9, 1), "sequence_ms": round(seq_qg, 3), "sequence_frac": frac(qg_bytes, seq_qg),
                               "api": "pointnet2_utils.QueryAndGroup, one launch per call"},
           "fill_probe_GB/s": round(fill_gbs, 1)}
    return out


def config_leg(name: str, extra: list, timeout_s: int = 120):
    """Another BASELINE.json configuration in a process of its own (as the `otf` / `b1` legs): this command with `extra`
    arguments and --leg (a short timed region, no legs / CPU baseline of its own), or tools/bench_backbone.py for configs[4].
    -> the compact record that goes into the main line."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    if name == "backbone16k":
        cmd = [sys.executable, os.path.join(here, "tools", "bench_backbone.py"), "--npoint", "2048", "512", "--steps", "60", "--warmup", "10"]
    else:
        cmd = [sys.executable, os.path.abspath(__file__), "--leg"] + extra
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s", "command": " ".join(cmd[1:])}
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        return {"error": (res.stderr or res.stdout)[-500:], "command": " ".join(cmd[1:])}
    d = json.loads(lines[-1])
    keep = ("metric", "value", "unit", "ms_per_step", "dtype", "roofline", "kernel_ms_per_step", "timed_blocks", "one_graph", "l1_stream", "pose_match")
    out = {k: d[k] for k in keep if k in d}
    # compact (the main line has to fit the driver's 8 KB tail): what the leg ran is its command; the numbers stay
    out.pop("config", None)
    if isinstance(out.get("timed_blocks"), dict):
        out["timed_blocks"] = {k: out["timed_blocks"][k] for k in ("n", "ms_per_step_min", "ms_per_step_max") if k in out["timed_blocks"]}
    if isinstance(out.get("roofline"), dict):
        # (bound "mfma", unit TFLOP/s as the main line's roofline: not repeated per leg -- the line has to fit 8 KB)
        out["roofline"] = {k: out["roofline"][k] for k in ("achieved", "peak", "frac", "traffic", "traffic_measured", "avg_launch_us", "flops_per_launch", "share_of_kernel_time") if k in out["roofline"]}
    if isinstance(out.get("pose_match"), dict):
        out["pose_match"] = {k: out["pose_match"][k] for k in ("trajectories", "max_abs_dR", "max_abs_dt", "max_abs_ds", "agree_5deg5cm", "within_1e-4") if k in out["pose_match"]}
    if isinstance(out.get("kernel_ms_per_step"), dict):
        top = sorted(((k, v) for k, v in out["kernel_ms_per_step"].items() if not k.startswith("_")), key=lambda kv: -kv[1])[:4]
        out["kernel_ms_per_step"] = dict(top, _sum=out["kernel_ms_per_step"].get("_sum_captra_kernels"))
    out.pop("metric", None)
    if out.get("unit") == "frames/s":
        out.pop("unit")                 # (as the main line; backbone16k keeps its clouds/s)
    out["command"] = " ".join(os.path.relpath(c, here) if c.startswith(here) else c for c in cmd[1:])
    out["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def pmc_traffic(kernel_prefixes, suffix: str = "", stem: str = "bench"):
    """HBM bytes per launch of the MFMA family from the newest committed PMC summary (profiles/*_bench_pmc.json,
    written by tools/profile_round.sh from separate rocprofv3 --pmc passes): 2 x FETCH_SIZE (gfx950 correction,
    MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes, averaged over the family's launches.  The file carries the fingerprint
    of the kernel sources it was profiled on (`csrc_sha16`, tools/pmc_summary.py); a file taken on other sources is NOT
    used: -> (None, name, reason)."""
    import glob
    # suffix: the counter file of another configuration of the same command (tools/profile_round.sh <tag> _bf16 --mlp-dtype bf16
    # writes profiles/<tag>_bench_bf16_pmc.json)
    # stem: "backbone16k" for tools/profile_backbone.sh's file (profiles/<tag>_backbone16k_pmc.json)
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"*_{stem}{suffix}_pmc.json")))
    if not files:
        return None, None, f"no profiles/*_{stem}{suffix}_pmc.json"
    with open(files[-1]) as fh:
        data = json.load(fh)
    name = os.path.basename(files[-1])
    stamp, now = data.get("csrc_sha16"), csrc_fingerprint()
    if stamp != now:
        return None, name, (f"{name} was profiled on kernel sources {stamp or '(unstamped)'}, this tree is {now}: re-run "
                            "tools/profile_round.sh")
    tot, n = 0.0, 0
    for kname, c in data.get("kernels", {}).items():
        if kname.startswith(tuple(kernel_prefixes)) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            k = min(c["FETCH_SIZE"]["dispatches"], c["WRITE_SIZE"]["dispatches"])
            tot += k * 1024.0 * (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"])
            n += k
    return (tot / n if n else None), name, None


def live_pmc_traffic(kernel_prefixes, extra_args, timeout_s: int = 150):
    """HBM bytes per launch of the MFMA family MEASURED IN THIS RUN: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE; counters
    only, with --kernel-trace -- the combination MI355X_MICROARCH.md prescribes) over a child of this script running 2 eager steps
    of the same workload, parsed like tools/pmc_summary.py, corrected like pmc_traffic (2 x FETCH_SIZE on gfx950).
    -> (bytes or None, note).  Any failure (no rocprofv3, a pass that times out or exits non-zero, nothing parsed) returns None and
    the caller keeps the committed profile's figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself under a profiler"
    sys.path.insert(0, os.path.join(here, "tools"))
    try:
        from pmc_summary import short
    finally:
        sys.path.pop(0)
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--repeats", "1", "--min-timed-s", "0", "--min-warmup", "1",
             "--no-pose-match", "--no-cpu-baseline", "--no-kernel-timing", "--no-otf", "--no-b1", "--no-legs", "--no-graph", "--no-overlap",
             "--no-live-traffic"] + list(extra_args)
    table = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(td, counter)
                res = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--"] + child,
                                     cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                if res.returncode != 0:
                    return None, f"rocprofv3 --pmc {counter} exited {res.returncode}"
                per = {}
                for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] != counter:
                                continue
                            cell = per.setdefault(row["Dispatch_Id"], [short(row["Kernel_Name"]), 0.0])
                            cell[1] += float(row["Counter_Value"])
                for name, v in per.values():
                    c = table.setdefault(name, {}).setdefault(counter, [0.0, 0])
                    c[0] += v
                    c[1] += 1
    except Exception as e:          # (a timeout, a parse error: the committed figure stays)
        return None, f"{type(e).__name__}: {e}"
    tot, n = 0.0, 0
    for kname, c in table.items():
        if kname.startswith(tuple(kernel_prefixes)) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            k = min(c["FETCH_SIZE"][1], c["WRITE_SIZE"][1])
            tot += k * 1024.0 * (2.0 * c["FETCH_SIZE"][0] / c["FETCH_SIZE"][1] + c["WRITE_SIZE"][0] / c["WRITE_SIZE"][1])
            n += k
    if not n:
        return None, "no dispatch of the family in the counter passes"
    return tot / n, f"{n} dispatches"


def pose_match(cfg, sd, frame_cpu, prev_pose, new_pose, which=(0, 1)):
    """Accuracy of the timed trajectories themselves (outside the timed region): the LAST step of the run, for trajectories
    `which` of this rank, redone by the CPU oracle (oracle/model.py track_step: C geometry + torch-CPU shared MLPs) from the
    same previous pose -> max |dR|, |dt|, |ds| and 5 deg / 5 cm agreement (eval_part_model, part_dof_utils.py:54-67)."""
    from captra_amd.pose_utils.part_dof_utils import eval_part_model
    from oracle import model as OM
    idx = list(which)
    prev = {k: v[idx].float().cpu().numpy() for k, v in prev_pose.items()}
    ours = {k: v[idx].float().cpu() for k, v in new_pose.items()}
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    ref, _ = OM.track_step(sd, cfg, frame_cpu["points"][idx].numpy(), frame_cpu["meta"]["points_mean"][idx].numpy(), prev, "torch")
    ref = {k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}
    d = eval_part_model(ref, ours, yaxis_only=bool(cfg["obj_sym"]))
    hit = torch.logical_and(d["rdiff"] <= 5.0, d["tdiff"] <= 0.05).float().mean()
    return {"vs": "oracle/model.py track_step (CPU) on the last timed frame, same previous pose", "trajectories": idx,
            "max_abs_dR": float((ref["rotation"] - ours["rotation"]).abs().max()),
            "max_abs_dt": float((ref["translation"] - ours["translation"]).abs().max()),
            "max_abs_ds": float((ref["scale"] - ours["scale"]).abs().max()),
            "rdiff_deg_max": float(d["rdiff"].max()), "tdiff_m_max": float(d["tdiff"].max()),
            "agree_5deg5cm": float(hit), "within_1e-4": bool(max(float((ref[k] - ours[k]).abs().max()) for k in ref) <= 1e-4)}


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_command(n: int, port: int | None = None) -> list:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else _free_port()), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]


def launch_plan(n: int) -> dict:
    """What `python bench.py --gpus n` would do, without doing it (no GPU is touched): the launcher command, one rank per
    GPU (LOCAL_RANK r -> cuda:r), RCCL ("nccl") for the per-frame pose all-gather, 32 trajectories per rank."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    share = os.environ.get("CAPTRA_BENCH_SHARE_GPU") == "1"
    return {"gpus_requested": n, "gpus_visible": have,
            "would_refuse": bool(n > 1 and have < n and not share),
            "command": "python bench.py (this process, no launcher)" if n == 1 else " ".join(spawn_command(n, port=29500)) + "   # port: a free one is picked at launch",
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")},
            "cpu_binding": "each rank pins its host threads to its share of the cores NUMA-local to its GPU (sysfs local_cpulist; an even split of the allowed cores when unknown)",
            "ranks": [{"rank": r, "local_rank": r, "device": f"cuda:{r % have if share and have else r}",
                       # what this rank tracks: the --category of the command line, or (mix6) NOCS category 1 + rank mod 6
                       "category": (WORKLOADS[MIX6[r % 6]][0] if "mix6" in sys.argv else
                                    WORKLOADS[sys.argv[sys.argv.index("--category") + 1]][0] if "--category" in sys.argv[:-1] and sys.argv[sys.argv.index("--category") + 1] in WORKLOADS else "1"),
                       "tracks": "32 trajectories (--batch), seeds (10 + rank) * 100 + b; category per --category (mix6: NOCS category 1 + rank mod 6)"}
                      for r in range(n)],
            "collective": "none (single rank)" if n == 1 else ("gloo (CAPTRA_BENCH_SHARE_GPU functional mode)" if share else
                          "nccl = RCCL: one async all_gather_into_tensor of the (32,P,14) fp32 pose records per step, waited for one step later; "
                          "barrier + device synchronize around every timed block"),
            "scaling": "weak"}


def self_spawn(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) of this very command under
    torch.distributed.run on 127.0.0.1 and return its exit code.  Refuses when the node has fewer than N GPUs (unless
    CAPTRA_BENCH_SHARE_GPU=1, the functional test mode in which ranks share devices over gloo)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("CAPTRA_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {n}: this node has {have} GPU(s); one rank per GPU is the only measured configuration")
    return subprocess.run(spawn_command(n)).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps before the timed region; at least 20 are run whatever the flag (SURVEY.md 8d)")
    ap.add_argument("--min-warmup", type=int, default=20, help=argparse.SUPPRESS)   # profiling recipes only (short traces)
    ap.add_argument("--repeats", type=int, default=10,
                    help="the timed block of exactly --steps steps (barrier + synchronize on both sides) is run this many times back to "
                         "back; `value` / `ms_per_step` come from the MEDIAN block, min / max are reported beside it")
    ap.add_argument("--min-timed-s", type=float, default=10.0,
                    help="keep running timed blocks (beyond --repeats) until they add up to this many seconds of GPU work")
    ap.add_argument("--batch", type=int, default=32, help="trajectories per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-pose-match", action="store_true", help="skip the CPU-oracle check of the timed trajectories' last step")
    ap.add_argument("--no-otf", action="store_true", help="skip the `otf` leg (the EvalTrackModel loop with the on-the-fly re-crop)")
    ap.add_argument("--otf-only", action="store_true", help=argparse.SUPPRESS)          # the `otf` leg's own process: prints that object only
    ap.add_argument("--b1-only", action="store_true", help=argparse.SUPPRESS)           # the `b1` leg's own process
    ap.add_argument("--no-b1", action="store_true", help="skip the `b1` leg (single-trajectory latency, pre-cropped and nocs_otf)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 counter passes over a 2-step child, ~40 s): keep the committed profile's figure")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the `bf16` (BASELINE.json configs[2]'s arithmetic), `drawers` (configs[3]) and `backbone16k` (configs[4]) legs")
    ap.add_argument("--leg", action="store_true", help=argparse.SUPPRESS)               # a configuration leg's own process: short timed region, the line only
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--lanes", type=int, default=0,
                    help="sub-batches of the rank's trajectories, each replaying its own captured step on its own stream, free-running "
                         "(captra_amd.graph.TrackLanes); 1 = one graph for the whole batch; 0 = 2 from 32 trajectories per GPU on "
                         "(measured: +2.8 %% at 32 and 64, -1 %% at 8 and 16)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run CoordinateNet and RotationNet one after the other instead of side by side on two streams (what the "
                         "per-kernel timing pass and the rocprofv3 recipes use: isolated kernel durations)")
    ap.add_argument("--mlp-dtype", default="fp32", choices=["fp32", "bf16", "f32x6"],
                    help="fp32 = the metric's configuration (exact); bf16 = bf16 MFMA operands / fp32 accumulation for the shared "
                         "MLPs (BASELINE.json configs[2]'s arithmetic) -- reported with dtype \"bf16\", not the headline; f32x6 = three-way "
                         "bf16 split of both operands, six bf16 MFMAs per k-step, fp32 accumulation (fp32-equivalent: <= 2e-6 of the exact "
                         "chain per layer) -- reported with dtype \"f32x6\", not the headline")
    ap.add_argument("--category", default="bottle", choices=sorted(WORKLOADS) + ["mix6"],
                    help="bottle = BASELINE.json configs[1] (the metric's configuration); the other object classes; mix6 = "
                         "BASELINE.json configs[2]'s serving mix: rank r tracks NOCS category 1 + r mod 6 with that category's weights")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the launch plan of `--gpus N` as one JSON object (the exact torch.distributed.run command the self-spawn "
                         "would run, the rank -> device map, the collective backend, what each rank tracks) and exit without touching a GPU")
    ap.add_argument("--no-pw-pair", action="store_true", help=argparse.SUPPRESS)       # A/B: dense layers without paired column tiles
    ap.add_argument("--pw-occ", type=int, default=0, help=argparse.SUPPRESS)           # A/B: dense layers' workgroups per CU (2 / 3)
    args = ap.parse_args()
    if args.dry_run:
        print(json.dumps(launch_plan(args.gpus)))
        return
    if args.leg:
        args.no_cpu_baseline = args.no_otf = args.no_b1 = args.no_legs = True
        args.no_pose_match = args.mlp_dtype != "f32x6"          # (the f32x6 leg carries its own pose_match: the mode's claim is accuracy + speed)
        args.min_timed_s = min(args.min_timed_s, 2.0)
        args.repeats = min(args.repeats, 5)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.otf_only:
        print(json.dumps(otf_leg(args.batch, torch.device("cuda", 0))))
        return
    if args.b1_only:
        print(json.dumps(b1_leg(torch.device("cuda", 0))))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_spawn(args.gpus))        # one rank per GPU, rendezvous on 127.0.0.1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    if args.no_pw_pair or args.pw_occ:
        from captra_amd import _lib as _knobs
        if args.pw_occ:
            _knobs.lib().captra_pw_set_occupancy(ctypes.c_int(args.pw_occ))
        if args.no_pw_pair:
            _knobs.lib().captra_pw_set_pair(ctypes.c_int(0))
    if args.category == "mix6":
        args.category = MIX6[rank % 6]
        args.mix6 = True
    share = os.environ.get("CAPTRA_BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    if world > ndev and not share:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) on this node")
    dev_index = local_rank % ndev if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = None
    force_dist = os.environ.get("CAPTRA_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ   # world 1 through RCCL too (1-GPU boxes)
    if world > 1 or force_dist:
        import torch.distributed as dist
        backend = "gloo" if share else "nccl"        # "nccl" IS RCCL on ROCm; gloo only in the shared-GPU functional test mode
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    from captra_amd import _lib, fused
    from captra_amd.parallel import PoseExchange, bind_rank_cpus
    bound_cpus = bind_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), dev_index) if world > 1 else None

    cfg, sd, model, data = build_workload(args.batch, device, category=args.category, traj_seed=rank, mlp_dtype=args.mlp_dtype)
    B, P = args.batch, cfg["num_parts"]
    if args.lanes == 0:
        args.lanes = 2 if B >= 32 and B % 2 == 0 else 1
    if args.no_overlap or args.no_graph:
        args.lanes = 1
    exchange = PoseExchange(B, P, device, world, rank, collective=dist is not None)
    nframes = len(model.feed_dict)
    pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}

    model.overlap_nets = not args.no_overlap
    use_graph = not args.no_graph
    graph = lanes = None
    if use_graph:
        from captra_amd.graph import TrackLanes, TrackStepGraph
        f1 = model.feed_dict[1]
        if args.lanes > 1 and B % args.lanes == 0:
            graph = lanes = TrackLanes(model, f1["points"], f1["points_mean"], pose, lanes=args.lanes)
        else:
            graph = TrackStepGraph(model, f1["points"], f1["points_mean"], pose)

    def frame_of(i):
        return 1 + i % (nframes - 1)

    def step(i, pose, timing_pass=False):
        f = frame_of(i)
        if graph is not None and not timing_pass:
            fd = model.feed_dict[f]
            if lanes is not None:
                # every lane hands its pose over to itself and runs ahead; this stream only waits (on the GPU) for the
                # frame's records and feeds them to the exchange
                new_pose = lanes.gather(lanes.step(fd["points"], fd["points_mean"]))
            else:
                new_pose = graph.replay(fd["points"], fd["points_mean"], pose)
        else:
            with torch.no_grad():
                _, new_pose = model.track_step(model.feed_dict[f], model.npcs_feed_dict[f], pose)
        # every rank ends up holding all poses of the frame; the gather of step i completes under step i + 1's kernels
        # (no rank needs remote poses to proceed) -- the wait below is for step i - 1's
        exchange.wait()
        exchange.all_gather(new_pose, async_op=True)
        return new_pose

    warm = max(args.warmup, args.min_warmup)
    for i in range(warm):
        pose = step(i, pose)

    def sync():
        exchange.wait()                           # the last step's gather belongs to the block it was issued in
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    timing = not args.no_kernel_timing
    eager_timing = timing and graph is None       # events can bracket kernels only when they are launched eagerly
    done = warm
    blocks = []                                   # seconds per timed block of exactly args.steps steps, this rank
    r = -1
    while True:
        r += 1
        # at least --repeats blocks, and blocks until the timed region adds up to --min-timed-s (the GPU stays busy long enough
        # for an outside sampler to see it; the value is still the MEDIAN block of exactly --steps steps)
        if r >= max(args.repeats, 1) and (sum(blocks) >= args.min_timed_s or r >= 2000):
            break
        sync()
        if eager_timing and r == 0:
            _lib.prof_reset()
            _lib.prof_enable(True)
            fused.work_reset(True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            pose = step(done + i, pose)
        sync()
        blocks.append(time.perf_counter() - t0)
        done += args.steps
        if eager_timing and r == 0:
            _lib.prof_enable(False)
            fused.WORK["on"] = False
    # the last step once more, keeping its input and output pose, for pose_match (outside every timed block)
    if lanes is not None:
        prev_pose = {k: v.clone() for k, v in pose.items()}
        last_pose = {k: v.clone() for k, v in step(done, pose).items()}
    else:
        prev_pose = {k: v.clone() for k, v in pose.items()}
        last_pose = {k: v.clone() for k, v in step(done, prev_pose).items()}
    last_frame = frame_of(done)
    torch.cuda.synchronize()
    # the level-1 stream kernel's own stamps of the step just replayed (100 MHz counter): where the sampler ended inside the launch
    l1_spans = None
    if getattr(model, "_l1_scratch", None) is not None:
        model.check_l1_stream()
        sp = fused.sa1_stream_spans(model._l1_scratch)
        l1_spans = {"sampler_us": round(sp[0], 1), "launch_us": round(sp[1], 1)}
    timed_steps = args.steps
    if timing and not eager_timing:
        # the timed blocks replayed a hipGraph; the per-kernel HIP events come from the same steps
        # launched eagerly right after them (same kernels, same shapes, same stream)
        p2 = {k: v.clone() for k, v in last_pose.items()}
        model.overlap_nets = False   # kernels one at a time: a launch's duration is the kernel's own, not its share of a busy chip
        _lib.prof_reset()
        _lib.prof_enable(True)
        fused.work_reset(True)
        for i in range(timed_steps):
            p2 = step(done + 1 + i, p2, timing_pass=True)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        fused.WORK["on"] = False
    rccl_world = 1
    per_rank_ms = [[1e3 * b / args.steps for b in blocks]]
    if dist is not None:
        # the world size as the collective library sees it: an actual all-gather of one word per rank
        mine = torch.tensor([rank], dtype=torch.int64, device=device)
        seen = torch.empty(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(seen, mine)
        rccl_world = int(seen.unique().numel())
        assert rccl_world == dist.get_world_size() == world, (rccl_world, dist.get_world_size(), world)
        mat = torch.tensor(blocks, dtype=torch.float64, device=device)
        allb = torch.empty(world * len(blocks), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allb, mat)
        allb = allb.view(world, len(blocks))
        per_rank_ms = (1e3 * allb / args.steps).tolist()
        blocks = allb.max(dim=0)[0].tolist()              # a block takes as long as its slowest rank
    assert all(torch.isfinite(v).all() for v in last_pose.values()), "non-finite pose"

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    order = sorted(blocks)
    elapsed = order[(len(order) - 1) // 2]                # median block (lower median for an even count)
    frames = B * world * args.steps
    out = {
        "metric": "tracked frames/sec (4096-pt clouds)", "value": round(frames / elapsed, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "f32x6": "f32x6"}[args.mlp_dtype],
        "data": "synthetic",
        "config": {"workload": f"{WORKLOADS[args.category][3]}, 4096 pts/frame, batch={B} trajectories per GPU, "
                               + ("fp32" if args.mlp_dtype == "fp32" else "f32x6: three-way bf16 split operands, six bf16 MFMAs per k-step, fp32 accumulation in the SA scales and the rotation heads' dense layers, exact fp32 elsewhere (opt-in; NOT the metric's configuration)" if args.mlp_dtype == "f32x6" else "bf16 MFMA operands / fp32 accumulation in the shared MLPs (BASELINE.json configs[2]'s arithmetic; NOT the metric's configuration)")
                               + (" (BASELINE.json configs[2]'s mix: rank r serves NOCS category 1 + r mod 6; rank 0's workload named here)" if getattr(args, "mix6", False)
                                  else " (BASELINE.json configs[1])" if args.category == "bottle" and args.mlp_dtype == "fp32" else " (BASELINE.json configs[3]: drawers)" if args.category == "drawers" else ""),
                   "points": 4096, "trajectories_per_gpu": B, "distinct_clouds_per_gpu": B,
                   "parallelism": f"dp{world} (trajectory-sharded, RCCL all-gather of poses)",
                   "weights": f"seeded default_rng(7), real architecture ({sum(v.numel() for v in sd.values()) / 1e6:.2f} M params), physical-regime plants",
                   "launch": (f"{args.lanes} free-running lanes of {B // args.lanes} trajectories, each a hipGraph replay of the step on its own stream" if lanes is not None
                              else "hipGraph replay of the step" if graph is not None else "eager launches")
                             + (", the two networks side by side" if not args.no_overlap else "")},
        "timed_blocks": {"n": len(blocks), "steps_each": args.steps, "ms_per_step_median": round(1e3 * elapsed / args.steps, 3),
                         "ms_per_step_min": round(1e3 * order[0] / args.steps, 3), "ms_per_step_max": round(1e3 * order[-1] / args.steps, 3),
                         "value_from": "median block of exactly `steps` steps (barrier + synchronize on both sides, max over ranks)",
                         "warmup_steps_run": warm,
                         # which hardware queues the lanes' streams were given is decided when the process creates them (DESIGN.md
                         # section 5); a bad draw shows as a second mode of the block times, so the distribution is reported
                         "ms_per_step_p10": round(1e3 * order[len(order) // 10] / args.steps, 3),
                         "ms_per_step_p90": round(1e3 * order[(9 * len(order)) // 10 if len(order) > 1 else 0] / args.steps, 3),
                         "bimodal": bool(len(order) >= 5 and order[(9 * len(order)) // 10] > 1.05 * order[len(order) // 10])},
        "rccl_world_size": rccl_world,
        "l1_stream": l1_spans,
        "rank0_cpu_affinity": (f"{len(bound_cpus)} cores next to its GPU ({bound_cpus[0]}..{bound_cpus[-1]}); every rank binds to its own share "
                               "(captra_amd.parallel.bind_rank_cpus)") if bound_cpus else "unbound (single rank)",
        "collective_backend": ("none (single rank)" if dist is None else "nccl (RCCL)" if backend == "nccl" else
                               "gloo -- CAPTRA_BENCH_SHARE_GPU functional test mode: ranks share GPUs, NOT a scaling measurement"),
        # (every block of every rank only when there is more than one rank: at world 1 it is `timed_blocks`)
        "per_rank_ms_per_step": [[round(x, 3) for x in row] for row in per_rank_ms] if world > 1 else [[round(sorted(per_rank_ms[0])[len(per_rank_ms[0]) // 2], 3)]],
    }
    if timing:
        fams = {}
        for name in _lib.prof_names():
            ms, n = _lib.prof_read(name)
            if n:
                fams[name] = {"ms_total": ms, "launches": n}
        total_ms = sum(v["ms_total"] for v in fams.values())
        # the fp32-MFMA shared-MLP kernels: the fused SA scale (sa_fused.hip) and the layer kernel template
        # (pointwise_mlp.hip: pointwise_mlp / sa_group_mlp / mlp_max entry points)
        # (the level-1 stream kernel's MLPs are timed with its sampler -- a latency-bound launch -- and stay out of the family)
        mlp = ["sa_scale_fused", "pointwise_mlp", "mlp_chain3", "coord_tail", "sa_group_mlp", "mlp_max", "neck_chain"]
        if args.mlp_dtype == "f32x6":
            # the mode's own kernels (csrc/sa_x6.hip, csrc/dense_x6.hip) against the bf16 peak / 6: six bf16 MFMAs per fp32-equivalent product
            mlp = ["sa_scale_x6", "pointwise_mlp_x6", "mlp_chain3_x6", "coord_tail_x6"]
        mlp_ms = sum(fams[k]["ms_total"] for k in mlp if k in fams)
        mlp_launches = sum(fams[k]["launches"] for k in mlp if k in fams)
        mlp_flops = sum(fused.WORK["flops"].get(k, 0.0) for k in mlp)
        dominant = max(fams, key=lambda k: fams[k]["ms_total"]) if fams else None
        if mlp_ms > 0:
            ach = mlp_flops / (mlp_ms * 1e-3) / 1e12
            peak = PEAK_F32_MFMA_TFLOPS if args.mlp_dtype == "fp32" else round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1) if args.mlp_dtype == "f32x6" else PEAK_BF16_MFMA_TFLOPS
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": None,
                               "kernel": ("fp32 MFMA shared-MLP family: sa_wave_pipe_kernel (SA2 scales, dominant) + sa_wave_lds_kernel (SA1) + mlp_chain3_kernel + coord_tail_kernel + pw_direct_kernel"
                                          if args.mlp_dtype == "fp32" else "f32x6 family: sa_x6_kernel (SA1 / SA2 scales) + dense_x6_kernel (rotation heads' 128 -> 512 -> 512 -> 256) + chain_x6_kernel (FP1 + conv1, CoordinateNet tail); fp32-equivalent flops against the bf16 MFMA peak / 6; the layers left on the exact fp32 kernels are not counted" if args.mlp_dtype == "f32x6" else "bf16 MFMA family: sa2_bf16_kernel / sa_bf16_kernel + tb_head12p_kernel / tb_layer_kernel + chain_bf16_kernel + pw_bf16_kernel (the level-1 stream kernel's MLPs are timed with its sampler and not counted here)"),
                               "avg_launch_us": round(1e3 * mlp_ms / max(mlp_launches, 1), 2),
                               "flops_per_launch": round(mlp_flops / max(mlp_launches, 1)),
                               "share_of_kernel_time": round(mlp_ms / max(total_ms, 1e-9), 3), "dominant_family": dominant}
            # the counter file of THIS configuration (tools/profile_round.sh <tag> <suffix> ...): bf16 / drawers have their own
            sfx = ("_f32x6" if args.mlp_dtype == "f32x6" else "_bf16" if args.mlp_dtype != "fp32" else "") + ("_drawers" if args.category == "drawers" else "")
            mlp_prefixes = (["sa_x6_kernel", "dense_x6_kernel", "chain_x6_kernel"] if args.mlp_dtype == "f32x6" else ["sa_bf16_kernel", "sa2_bf16_kernel", "tb_layer_kernel", "tb_head12_kernel", "tb_head12p_kernel", "neck_chain_kernel", "chain_bf16_kernel", "pw_bf16pm_kernel", "pw_bf16pm_affs_kernel", "pw_bf16_kernel"]
                            if args.mlp_dtype != "fp32" else
                            ["sa_wave_kernel", "sa_wave_lds_kernel", "sa_wave_pipe_kernel", "sa_fused_kernel", "mlp_chain3_kernel", "coord_tail_kernel", "pw_direct_kernel", "pw_direct_max_kernel", "pw_mlp_kernel"])
            traffic, src, why = pmc_traffic(mlp_prefixes, sfx)
            if traffic is not None:
                out["roofline"]["traffic"] = round(traffic)
                out["roofline"]["traffic_source"] = f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per launch, kernel sources {csrc_fingerprint()} = this tree)"
            else:
                out["roofline"]["traffic_source"] = f"null: {why}"
                print(f"bench.py: roofline.traffic dropped -- {why}", file=sys.stderr)
            out["_traffic_family"] = (mlp_prefixes, sfx)
        if "ball_query" in fams:
            # the scan is N x M distance tests per cloud (VALU work: 8 packed-fp32 instructions per 128 tests + the ordered
            # compaction), not a byte stream: reported as pair tests per second against the packed-fp32 VALU rate, with the
            # section 8(d) bytes beside it for the record
            bq = fams["ball_query"]
            nbytes = fused.WORK["bytes"].get("ball_query", 0.0)
            pairs = fused.WORK["flops"].get("ball_query", 0.0)
            sec = bq["ms_total"] * 1e-3
            peak_pairs = 256 * 4 * 2.4e9 * 64 / 8.0              # one wave-instruction per 2 cycles per SIMD, 8 instructions per 64 lanes x 2 tests ... upper bound
            # (the scan leaves a centre when every radius' list is full -- count the pairs it makes, not N x M)
            try:
                share = ball_query_scanned_fraction(data[last_frame]["points"].to(device), cfg["pointnet"]["camera"])
            except Exception:                                   # noqa: BLE001  (measurement only: never fails the line)
                share = None
            pairs_nm = pairs
            pairs = pairs * share if share else pairs
            out["roofline_ball_query"] = {"bound": "valu", "unit": "pair tests/s", "achieved": round(pairs / sec) if pairs else None,
                                          "peak": round(peak_pairs), "frac": round(pairs / sec / peak_pairs, 4) if pairs else None,
                                          "scanned_share_of_NxM": round(share, 4) if share else None, "NxM_per_s": round(pairs_nm / sec) if pairs_nm else None,
                                          "algorithmic_GB/s": round(nbytes / sec / 1e9, 1),
                                          "avg_launch_us": round(1e3 * bq["ms_total"] / bq["launches"], 2)}
        out["kernel_ms_per_step"] = {k: round(v["ms_total"] / timed_steps, 3) for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms_total"])}
        out["kernel_ms_per_step"]["_sum_captra_kernels"] = round(total_ms / timed_steps, 3)
    if world == 1 and timing:
        out["hbm_ops"] = hbm_ops_roofline(B, device)
        bq_ms = out.get("kernel_ms_per_step", {}).get("ball_query")
        if bq_ms:
            # what the timed step spends on the same job: the ball-query launches only -- grouping happens inside the SA
            # kernels' operand loads and moves none of the 4*C*M*K bytes
            out["hbm_ops"]["product_path_ball_query_ms_per_step"] = bq_ms
        if "roofline" in out:
            # north_star's ">= 60 % of the HBM roofline on ball_query + group", where the driver's parser keeps it: through the reference's
            # per-op signatures (per-launch brackets / one bracket), through the per-level batched entries, and through the reference's
            # QueryAndGroup module (one launch per call)
            h = out["hbm_ops"]
            out["roofline"]["hbm_ops"] = {"frac": h["frac"], "sequence_frac": h["sequence"]["frac"], "per_level_frac": h["per_level"]["frac"],
                                          "query_and_group_frac": h["query_and_group"]["frac"], "query_and_group_sequence_frac": h["query_and_group"]["sequence_frac"],
                                          "peak_GB/s": PEAK_HBM_GBS}
    if not args.no_pose_match and args.mlp_dtype in ("fp32", "f32x6"):
        # one trajectory of EACH lane (lanes are contiguous halves of the batch): the first and the last
        out["pose_match"] = pose_match(cfg, sd, data[last_frame], prev_pose, last_pose, which=(0, B - 1) if B > 1 else (0,))
    if world == 1 and not args.no_otf and args.mlp_dtype == "fp32" and args.category == "bottle":
        # in a process of its own, as the tracker is run (`python -m captra_amd.track --nocs_otf True`): which hardware queues
        # the loop's streams get depends on how many streams the process has created before, and
        # this process has created the bench's
        import subprocess
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--otf-only", "--batch", str(B)], capture_output=True, text=True)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        out["otf"] = json.loads(lines[-1]) if res.returncode == 0 and lines else {"error": (res.stderr or res.stdout)[-500:]}
    if world == 1 and not args.no_b1 and args.mlp_dtype == "fp32" and args.category == "bottle":
        import subprocess
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--b1-only"], capture_output=True, text=True)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        out["b1"] = json.loads(lines[-1]) if res.returncode == 0 and lines else {"error": (res.stderr or res.stdout)[-500:]}
    if world == 1 and not args.no_legs and args.mlp_dtype == "fp32" and args.category == "bottle":
        # BASELINE.json configs[2] (arithmetic), [3] and [4], each measured by this run in a process of its own
        out["legs"] = {"f32x6": config_leg("f32x6", ["--mlp-dtype", "f32x6", "--batch", str(B)]),
                       "bf16": config_leg("bf16", ["--mlp-dtype", "bf16", "--batch", str(B)]),
                       "drawers": config_leg("drawers", ["--category", "drawers", "--batch", str(B)]),
                       "backbone16k": config_leg("backbone16k", [])}
    fam = out.pop("_traffic_family", None)
    if fam is not None and world == 1 and not args.no_live_traffic:
        # roofline.traffic measured in THIS run (two counter passes over a 2-step child of the same workload); the committed
        # profile's figure stays beside it, and stays the value if the passes cannot run here
        extra = ["--batch", str(B)] + (["--mlp-dtype", args.mlp_dtype] if args.mlp_dtype != "fp32" else []) + (["--category", args.category] if args.category != "bottle" else [])
        t0 = time.perf_counter()
        live, note = live_pmc_traffic(fam[0], extra)
        if live is not None:
            out["roofline"]["traffic_committed_profile"] = out["roofline"].get("traffic")
            out["roofline"]["traffic"] = round(live)
            if args.leg:
                out["roofline"]["traffic_measured"] = "in this run"     # (a leg's record keeps this flag instead of the source text)
            out["roofline"]["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a 2-step eager child of this workload, "
                                                 f"per launch of the family ({note}; {time.perf_counter() - t0:.0f} s)")
        else:
            out["roofline"]["traffic_live"] = f"not measured here ({note}): the committed profile's figure"
    if world == 1 and not args.no_cpu_baseline and args.category == "bottle":
        out["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_budget)
        out["cpu_baseline"]["reference_cpu_path_in_authoring_container"] = {
            "value": 3.08, "unit": "frames/s", "cores": 8,
            "note": "BASELINE.md section 2: recorded in the authoring container, not re-measured here"}
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
