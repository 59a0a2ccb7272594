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
import os
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
    parser.add_argument("--mlp_dtype", default="fp32", choices=["fp32", "bf16", "f32x6"],
                        help="bf16: bf16 MFMA operands / fp32 accumulation in the shared MLPs (opt-in; default exact fp32)")
    parser.add_argument("--hipgraph", action="store_true", default=False,
                        help="replay one captured hipGraph per frame (same kernels, no per-launch host overhead)")
    return parser.parse_args(argv)


class Ranks:
    """One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK from the launcher's environment
    (`python -m torch.distributed.run --nproc-per-node G -m captra_amd.track ...`); world 1 = the plain single-process harness.
    Trajectories shard over ranks (parallel.shard_range), every frame's pose records are all-gathered (RCCL over xGMI; gloo
    with CAPTRA_DIST_BACKEND=gloo), rank 0 prints and writes the result pickles of every rank's trajectories."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.owns_group = False

    def init(self, cfg):
        if self.world == 1:
            return
        import torch.distributed as dist
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        backend = os.environ.get("CAPTRA_DIST_BACKEND") or ("nccl" if ndev else "gloo")
        if ndev:
            if self.local_rank >= ndev and backend == "nccl":
                raise SystemExit(f"rank {self.rank}: LOCAL_RANK {self.local_rank} but {ndev} GPU(s); one rank per GPU")
            cfg["device"] = f"cuda:{self.local_rank % ndev}"        # (sharing a device is possible over gloo only: functional tests)
            torch.cuda.set_device(self.local_rank % ndev)
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device(cfg["device"]))
            else:
                dist.init_process_group(backend=backend)
            self.owns_group = True
        self.dist = dist

    def max_int(self, v: int, device) -> int:
        if self.dist is None:
            return v
        t = torch.tensor([v], dtype=torch.int64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def broadcast_object(self, obj):
        """rank 0's `obj` on every rank."""
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather_objects(self, obj):
        """-> list over ranks on rank 0 (None elsewhere)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.owns_group:
                self.dist.destroy_process_group()


class FramePoseGather:
    """The per-frame pose exchange of one batch: `model.frame_hook = gather` makes every frame's packed pose records
    [R(9) t(3) s(1) valid(1)] x P of all ranks' trajectories available on every rank (parallel.PoseExchange, async: the
    all-gather of frame i runs under the kernels of frame i+1; nobody needs remote poses to proceed).  A rank whose batch is
    short (or empty) contributes invalid records, so that every rank issues the same collectives."""

    def __init__(self, capacity: int, num_parts: int, device, ranks: Ranks, local_batch: int):
        from .parallel import PoseExchange
        self.ex = PoseExchange(capacity, num_parts, device, ranks.world, ranks.rank)
        self.capacity, self.P, self.b, self.device = capacity, num_parts, local_batch, device
        self.frames = []                   # gathered (world * capacity, P, 14) records, one per frame, on this rank
        self.issued = 0

    def _collect(self):
        if self.issued > len(self.frames):
            self.frames.append(self.ex.wait().clone())

    def __call__(self, i: int, pose):
        from .parallel import pack_pose
        self._collect()                    # frame i-1 has landed (stream-ordered wait) before its buffers are reused
        rec = torch.zeros(self.capacity, self.P, 14, dtype=torch.float32, device=self.device)
        if pose is not None and self.b:
            rec[:self.b] = pack_pose({k: v[:self.b] for k, v in pose.items()})
        self.ex.local.copy_(rec)
        self.ex.all_gather_packed(async_op=True)
        self.issued += 1

    def finish(self, total_frames: int):
        """Ranks with fewer frames in this batch (or none) issue the remaining all-gathers with invalid records."""
        while self.issued < total_frames:
            self(self.issued, None)
        self._collect()
        return self.frames


def _trajectory_sources(args, cfg):
    """The global, ordered list of trajectories: ('synthetic', kind, seed) entries or file names."""
    if args.data.startswith("synthetic"):
        kind = args.data.split(":")[1] if ":" in args.data else ("nocs" if cfg["num_parts"] == 1 else "arti")
        return [("synthetic", kind, args.seed + g) for g in range(args.num_traj)]
    files = sorted(glob.glob(pjoin(args.data, "*.npz")))
    if not files:
        raise SystemExit(f"no trajectory .npz files under {args.data}")
    return files


def _load_source(src, args):
    """One trajectory as a list over frames of batch-1 frame dicts.  A synthetic trajectory is generated from its own seed
    (batch 1), so its content does not depend on the rank or the batch it is tracked in."""
    if isinstance(src, tuple):
        try:
            from captra_amd import synthetic as clouds
        except ImportError as e:  # pragma: no cover
            raise SystemExit("--data synthetic needs the repository's tests/ package on sys.path") from e
        _, kind, seed = src
        return clouds.make_trajectory(kind, 1, args.num_frames, seed=seed)
    return stack_trajectories([load_trajectory_npz(src)])


def iter_batches(args, cfg, ranks: Ranks | None = None):
    """Yields lists over frames of batched frame dicts, at most cfg['batch_size'] trajectories of THIS rank's shard at a
    time (equal frame counts and cloud sizes batch together)."""
    from .parallel import shard_range
    from .trajectory_io import concat_frame_batches
    B = max(int(cfg.get("batch_size") or 1), 1)
    sources = _trajectory_sources(args, cfg)
    if ranks is not None and ranks.world > 1:
        sources = [sources[g] for g in shard_range(len(sources), ranks.world, ranks.rank)]
    trajs = [_load_source(s, args) for s in sources]
    if not args.data.startswith("synthetic"):
        trajs.sort(key=lambda t: len(t))                     # equal-length trajectories batch together
    i = 0
    while i < len(trajs):
        j = i
        shape = (len(trajs[i]), tuple(trajs[i][0]["points"].shape[1:]))
        while j < len(trajs) and j - i < B and (len(trajs[j]), tuple(trajs[j][0]["points"].shape[1:])) == shape:
            j += 1
        yield concat_frame_batches(trajs[i:j])
        i = j


def main(argv=None) -> dict:
    args = parse_args(argv)
    data_args = {k: getattr(args, k) for k in ("data", "num_traj", "num_frames", "random_init", "seed", "hipgraph", "mlp_dtype")}
    for k in data_args:
        delattr(args, k)
    cfg = get_config(args, save=False)
    cfg["hipgraph"] = data_args["hipgraph"]
    cfg["mlp_dtype"] = data_args["mlp_dtype"]          # per-model setting (EvalTrackModel.mlp_dtype), no process-wide switch
    args = argparse.Namespace(**vars(args), **data_args)
    ranks = Ranks()
    ranks.init(cfg)
    root = ranks.rank == 0

    log_dir = pjoin(cfg["experiment_dir"], "log")
    ensure_dirs(log_dir)
    logger = logging.getLogger("TestModel")
    logger.setLevel(logging.INFO)
    handler = logging.FileHandler(pjoin(log_dir, "log_test.txt" if ranks.world == 1 else f"log_test_rank{ranks.rank}.txt"))
    handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(handler)

    def log_string(msg):
        logger.info(msg)
        if root:
            print(msg)

    log_string("PARAMETER ...")
    log_string({k: v for k, v in cfg.items() if k not in ("obj", "pointnet")})
    if ranks.world > 1:
        log_string(f"rank {ranks.rank} of {ranks.world} on {cfg['device']}: trajectories shard over ranks, poses all-gathered per frame")

    trainer = Trainer(cfg, logger)
    if args.random_init:
        trainer.model.to(cfg["device"])
        log_string("random-initialised weights (--random_init): pose errors are meaningless, throughput only")
    else:
        trainer.resume()
    model = trainer.model
    device = cfg["device"]
    capacity = max(int(cfg.get("batch_size") or 1), 1)

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    test_loss = {"cnt": 0}
    time_dict = {"data_proc": 0.0, "network": 0.0}
    total_frames = 0
    zero_time = time.time()
    batches = iter_batches(args, cfg, ranks)
    i = 0
    while True:
        data = next(batches, None)
        # every rank takes part in every round of collectives: a rank that has run out of trajectories plays empty rounds
        if ranks.max_int(0 if data is None else 1, device) == 0:
            break
        b_local = 0 if data is None else len(data[0]["points"])
        t_local = 0 if data is None else len(data)
        num_frames = t_local * b_local
        total_frames += num_frames
        if root:
            print(f"Trajectory {i}, {num_frames:8} frames****************************")
        start_time = time.time()
        elapse = start_time - zero_time
        time_dict["data_proc"] += elapse
        if root:
            print(f"Data Preprocessing: {elapse:8.2f}s {num_frames / max(elapse, 1e-9):8.2f}FPS")
        gather, records = None, []
        if ranks.world > 1:
            # collectives are matched by ORDER: agree on the round's frame count before any rank starts its per-frame
            # all-gathers (a rank without trajectories issues them all in finish())
            t_all = ranks.max_int(t_local, device)
            gather = FramePoseGather(capacity, cfg["num_parts"], device, ranks, b_local)
            model.frame_hook = gather
            model.result_sink = records.extend          # rank 0 writes every rank's pickles (below)
        sync()
        start_time = time.time()
        loss_dict = {}
        if data is not None:
            pred_dict, loss_dict = trainer.test(data, save=cfg["save"], no_eval=cfg["no_eval"])
        if gather is not None:
            gathered = gather.finish(t_all)
            model.frame_hook = model.result_sink = None
        sync()
        elapse = time.time() - start_time
        time_dict["network"] += elapse
        if root:
            print(f"Network Forwarding: {elapse:8.2f}s {num_frames / max(elapse, 1e-9):8.2f}FPS")
        if gather is not None and cfg["save"]:
            _write_gathered_results(cfg, ranks, records, gathered, capacity)
        # per-trajectory averages (prediction and its initialisation); the per-frame tables stay in model.loss_dict
        flat = {f"{group}/{k}": float(v) * b_local for group in ("avg_pred", "avg_init")
                for k, v in (loss_dict.get(group) or {}).items() if _is_number(v)}
        flat["cnt"] = b_local if data is not None else 0
        add_dict(test_loss, flat)
        zero_time = time.time()
        i += 1

    # totals over ranks: frames add up, a phase takes as long as its slowest rank; losses are trajectory-weighted means
    all_stats = ranks.gather_objects({"frames": total_frames, "time": time_dict, "loss": test_loss})
    result = {"frames": total_frames, "network_s": time_dict["network"], "loss": {}}
    if root:
        frames = sum(s["frames"] for s in all_stats)
        tdict = {k: max(s["time"][k] for s in all_stats) for k in time_dict}
        print(f"Overall, {frames:8} frames****************************")
        print(f"Data Preprocessing: {tdict['data_proc']:8.2f}s {frames / max(tdict['data_proc'], 1e-9):8.2f}FPS")
        print(f"Network Forwarding: {tdict['network']:8.2f}s {frames / max(tdict['network'], 1e-9):8.2f}FPS")
        total = {}
        for s in all_stats:
            add_dict(total, s["loss"])
        cnt = max(total.pop("cnt", 0), 1)
        summary = {}
        for key, val in total.items():
            try:
                summary[key] = float(val) / cnt
            except (TypeError, ValueError):
                continue
            log_string("Test {} is {}".format(key, summary[key]))
        result = {"frames": frames, "network_s": tdict["network"], "loss": summary, "world": ranks.world}
    ranks.finish()
    logger.removeHandler(handler)
    handler.close()
    return result


def _write_gathered_results(cfg, ranks: Ranks, records, gathered, capacity: int) -> None:
    """Rank 0 writes the result pickles of every rank's trajectories of this batch.  The predicted poses in them are the
    ALL-GATHERED per-frame records (the bytes that crossed xGMI), checked against what the owning rank computed; corners,
    ground truth and frame numbers come with the rank's record (host objects, gathered once per batch)."""
    from .model import write_result_pickles
    from .parallel import unpack_pose
    per_rank = ranks.gather_objects(records)
    out, error = [], None
    if ranks.rank == 0:
        try:
            for r, recs in enumerate(per_rank):
                for j, (name, rec) in enumerate(recs):
                    for i, pose in enumerate(rec["pred"]["poses"]):
                        got, valid = unpack_pose(gathered[i][r * capacity + j:r * capacity + j + 1].cpu())
                        assert bool(valid.all()), f"rank {r} trajectory {j} frame {i}: record not marked valid"
                        for key in pose:
                            wire = got[key][0].reshape(pose[key].shape)
                            mine = torch.as_tensor(pose[key])
                            # bit patterns, not values: a NaN pose (diverged / empty part under --random_init) equals itself on the wire
                            if not torch.equal(wire.float().contiguous().view(torch.int32), mine.float().contiguous().view(torch.int32)):
                                raise RuntimeError(f"pose exchange of {name} frame {i} {key}: the all-gathered record differs from what rank {r} "
                                                   f"computed (max |diff| {float((wire.float() - mine.float()).abs().nan_to_num(0.0).max()):.3g}); "
                                                   "refusing to write a corrupted wire copy into the result pickles")
                            pose[key] = wire.clone() if torch.is_tensor(pose[key]) else wire.numpy().copy()
                    out.append((name, rec))
        except (RuntimeError, AssertionError) as e:
            error = str(e)
    # every rank learns the verdict and fails together: rank 0 alone raising would leave the others in their next collective until
    # the communicator's timeout
    error = ranks.broadcast_object(error)
    if error is not None:
        raise RuntimeError(error)
    if ranks.rank != 0:
        return
    write_result_pickles(cfg["experiment_dir"], out)


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
