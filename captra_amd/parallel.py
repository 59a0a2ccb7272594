"""Multi-GPU layer of the track loop: trajectory sharding + RCCL all-gather of per-frame poses.

The reference is single-process / single-GPU (`cuda:%d`, configs/config.py:68; no torch.distributed
anywhere).  The path shards over the batch of independent trajectories and nowhere
else (frame i needs pose i-1; a cloud is never split): trajectory b lives on rank b mod G with a
full copy of both nets' weights (15.8 MB), so the data path has NO collective.  The only exchange
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
