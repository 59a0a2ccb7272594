"""Multi-GPU layer of the track loop: trajectory sharding + RCCL all-gather of per-frame poses.

The reference is single-process / single-GPU (`cuda:%d`, configs/config.py:68; no torch.distributed
anywhere).  The path shards over the batch of independent trajectories and nowhere
else (frame i needs pose i-1; a cloud is never split): every rank owns a CONTIGUOUS, balanced range of the
trajectories (`shard_range`: the first `total % G` ranks hold one more) with a full copy of both nets'
weights (15.8 MB), so the data path has NO collective.  The only exchange
is the result: after each frame every rank contributes its packed pose records
[R(9) t(3) s(1) valid(1)] x P fp32 per trajectory and receives everyone's — 56·P bytes per
trajectory, latency-bound on xGMI.  One process per GPU, `torch.distributed`
backend "nccl" (= RCCL on ROCm) on the GPU box, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch

POSE_RECORD = 14  # floats per (trajectory, part): rotation 9, translation 3, scale 1, valid 1


def shard_range(total: int, world: int, rank: int) -> range:
    """Contiguous, balanced share of `total` trajectories for `rank` (first `total % world` ranks get one more)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' (the sysfs cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpu_slice(cpus: list[int], local_rank: int, local_world: int, sharers: list[int] | None = None) -> list[int]:
    """The host cores of one rank: `cpus` (the GPU's NUMA-local cores, or every allowed core) dealt evenly among the ranks that
    share them (`sharers`: the local ranks whose GPUs sit on the same NUMA node; None = all local ranks); never empty."""
    sharers = list(range(local_world)) if sharers is None else sorted(sharers)
    if local_rank not in sharers or not cpus:
        return list(cpus)
    per = max(len(cpus) // len(sharers), 1)
    i = sharers.index(local_rank)
    part = cpus[i * per:(i + 1) * per] if (i + 1) * per <= len(cpus) else cpus[-per:]
    return part or list(cpus)


def gpu_local_cpus(device_index: int) -> list[int] | None:
    """NUMA-local cores of GPU `device_index` from sysfs (/sys/bus/pci/devices/<bdf>/local_cpulist), None when unknown."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            cpus = parse_cpulist(fh.read())
        return cpus or None
    except Exception:
        return None


def bind_rank_cpus(local_rank: int, local_world: int, device_index: int | None = None) -> list[int] | None:
    """Pin this rank's host threads (the launch loop is single-threaded Python: one busy core per GPU) to its share of the
    cores next to its GPU, so that eight ranks do not migrate across sockets.  Returns the cores, or None when nothing was
    changed (one rank, or an OS without sched_setaffinity)."""
    import os
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    local = gpu_local_cpus(device_index) if device_index is not None else None
    sharers = None
    if local:
        local = [c for c in local if c in set(allowed)] or None
    if local:
        # the ranks whose GPUs report the same core list share it
        sharers = [r for r in range(local_world) if (gpu_local_cpus(r) or []) and set(gpu_local_cpus(r)) & set(local)]
    cores = cpu_slice(local or allowed, local_rank, local_world, sharers if local else None)
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        return None
    return cores


def pack_pose(pose: dict, valid: torch.Tensor | None = None) -> torch.Tensor:
    """{'rotation' (B,P,3,3), 'translation' (B,P,3,1), 'scale' (B,P)} -> (B,P,14) fp32."""
    B, P = pose["scale"].shape
    v = torch.ones(B, P, 1, dtype=torch.float32, device=pose["scale"].device) if valid is None else valid.float().reshape(B, P, 1)
    return torch.cat([pose["rotation"].reshape(B, P, 9).float(), pose["translation"].reshape(B, P, 3).float(),
                      pose["scale"].reshape(B, P, 1).float(), v], dim=-1).contiguous()


def unpack_pose(rec: torch.Tensor) -> tuple[dict, torch.Tensor]:
    """(B,P,14) -> (pose dict, valid (B,P) bool)."""
    B, P, _ = rec.shape
    pose = {"rotation": rec[..., 0:9].reshape(B, P, 3, 3), "translation": rec[..., 9:12].reshape(B, P, 3, 1),
            "scale": rec[..., 12]}
    return pose, rec[..., 13] > 0.5


class PoseExchange:
    """All-gather of the packed pose records of one frame.  Buffers are allocated once; with
    world == 1 it degenerates to a local copy so that the single-GPU step does the same packing."""

    def __init__(self, batch_per_rank: int, num_parts: int, device, world: int = 1, rank: int = 0, collective: bool | None = None):
        """collective: issue the all-gather through torch.distributed even at world 1 (None = only when world > 1): what a
        1-GPU box can exercise of the RCCL path."""
        self.world, self.rank = world, rank
        self.collective = world > 1 if collective is None else bool(collective)
        self.local = torch.empty(batch_per_rank, num_parts, POSE_RECORD, dtype=torch.float32, device=device)
        self.gathered = torch.empty(world * batch_per_rank, num_parts, POSE_RECORD, dtype=torch.float32, device=device)
        self.handle = None

    def all_gather(self, pose: dict, valid: torch.Tensor | None = None, async_op: bool = False) -> torch.Tensor:
        if self._pack_on_device(pose, valid):
            return self.gathered if not self.collective else self.all_gather_packed(async_op)
        self.local.copy_(pack_pose(pose, valid))
        return self.all_gather_packed(async_op)

    def _pack_on_device(self, pose: dict, valid) -> bool:
        """One launch for the packing (and, on a single rank without a collective, the "gathered" copy) instead of a cat, a
        fill and two or three copies per frame; False when the tensors are not what the kernel takes."""
        if not self.local.is_cuda:
            return False
        t = [pose["rotation"], pose["translation"], pose["scale"]] + ([] if valid is None else [valid])
        if not all(x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() for x in t):
            return False
        n = self.local.shape[0] * self.local.shape[1]
        # the kernel reads 9 n / 3 n / n / n floats unchecked: every tensor must have exactly that many, on the records' device
        if (pose["scale"].numel() != n or pose["rotation"].numel() != 9 * n or pose["translation"].numel() != 3 * n
                or (valid is not None and valid.numel() != n) or any(x.device != self.local.device for x in t)):
            return False
        from . import _lib as L
        with torch.cuda.device(self.local.device):
            L.call("captra_pack_pose", n, L.ptr(t[0]), L.ptr(t[1]), L.ptr(t[2]), None if valid is None else L.ptr(valid),
                   L.ptr(self.local), None if self.collective else L.ptr(self.gathered))
        return True

    def all_gather_packed(self, async_op: bool = False) -> torch.Tensor:
        """All-gather whatever `self.local` holds (records packed by the caller, e.g. a short batch padded with invalid ones)."""
        if not self.collective:
            self.gathered.copy_(self.local)
            return self.gathered
        import torch.distributed as dist
        self.handle = dist.all_gather_into_tensor(self.gathered, self.local, async_op=async_op)
        if not async_op:
            self.handle = None
        return self.gathered

    def wait(self) -> torch.Tensor:
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        return self.gathered


def allreduce_gradients(params, world: int, bucket_bytes: int = 32 << 20) -> None:
    """Data-parallel training step: average the gradients of `params` over the ranks, in flat buckets
    of ~bucket_bytes so that the 3.9 M parameters of a CAPTRA net (15.8 MB fp32) travel as ONE ring all-reduce over xGMI —
    per-tensor collectives would pay the ring's latency 190 times.  No-op for world == 1."""
    if world == 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    start = 0
    while start < len(grads):
        end, size = start, 0
        while end < len(grads) and (size == 0 or size + grads[end].numel() * grads[end].element_size() <= bucket_bytes):
            size += grads[end].numel() * grads[end].element_size()
            end += 1
        flat = torch.cat([g.reshape(-1) for g in grads[start:end]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        offset = 0
        for g in grads[start:end]:
            g.copy_(flat[offset:offset + g.numel()].view_as(g))
            offset += g.numel()
        start = end
