"""Operator layer with the reference's callables and dtypes (int32 indices, fp32 values).

Mirrors network/models/pointnet_lib/pointnet2_utils.py: `furthest_point_sample` (l.37),
`gather_operation` (l.76), `knn` (l.108), `three_nn` (l.141), `three_interpolate` (l.192),
`grouping_operation` (l.238), `ball_query` (l.271), plus the `QueryAndGroup` / `GroupAll`
modules.  Outputs are allocated here with `torch.empty(..., device=)` (the reference's legacy
`torch.cuda.IntTensor(...)` constructors allocate on the current device regardless of the
input's device) and handed to the caller-allocates C ABI through `pointnet2_cuda`.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import pointnet2_cuda as pointnet2


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.int32 else t.int().contiguous()


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) -> (B,npoint) int32; start index 0, lowest index wins ties."""
        xyz = _f32(xyz)
        B, N, _ = xyz.shape
        out = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""
        features, idx = _f32(features), _i32(idx)
        B, npoint = idx.shape
        _, C, N = features.shape
        out = torch.empty(B, C, npoint, dtype=torch.float32, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, out)
        ctx.for_backwards = (idx, C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.shape
        grad = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, _f32(grad_out), idx, grad)
        return grad, None


gather_operation = GatherOperation.apply


class KNN(Function):
    @staticmethod
    def forward(ctx, k: int, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,N,3), known (B,M,3) -> (sqrt distances (B,N,k), idx (B,N,k) int32)."""
        unknown, known = _f32(unknown), _f32(known)
        B, N, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.empty(B, N, k, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, k, dtype=torch.int32, device=unknown.device)
        pointnet2.knn_wrapper(B, N, m, k, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,N,3), known (B,M,3) -> (sqrt distances (B,N,3), idx (B,N,3) int32)."""
        unknown, known = _f32(unknown), _f32(known)
        B, N, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,C,M), idx/weight (B,n,3) -> (B,C,n)."""
        features, idx, weight = _f32(features), _i32(idx), _f32(weight)
        B, c, m = features.shape
        n = idx.shape[1]
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.shape
        grad = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, _f32(grad_out), idx, weight, grad)
        return grad, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
        features, idx = _f32(features), _i32(idx)
        B, npoint, nsample = idx.shape
        _, C, N = features.shape
        out = torch.empty(B, C, npoint, nsample, dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, npoint, nsample, features, idx, out)
        ctx.for_backwards = (idx, N)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.shape
        grad = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, _f32(grad_out), idx, grad)
        return grad, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32."""
        new_xyz, xyz = _f32(new_xyz), _f32(xyz)
        B, N, _ = xyz.shape
        npoint = new_xyz.shape[1]
        idx = torch.zeros(B, npoint, nsample, dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query + group (+ centre subtraction); features first, xyz last when both are used."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None) -> torch.Tensor:
        needs_grad = torch.is_grad_enabled() and (xyz.requires_grad or new_xyz.requires_grad or (features is not None and features.requires_grad))
        if xyz.is_cuda and not needs_grad:
            # inference: the whole module as ONE launch (captra_query_and_group) -- the lists never leave LDS, the cloud is staged
            # once for the search and for the coordinate channels; same values as the ops below, bit for bit
            from .. import fused
            out = fused.query_and_group(self.radius, self.nsample, xyz.contiguous(), new_xyz.contiguous(),
                                        None if features is None else features.contiguous(), self.use_xyz)
            if out is not None:
                return out
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped, grouped_xyz], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None) -> torch.Tensor:
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class KNNAndGroup(nn.Module):
    """k-nearest-neighbour grouping (reference pointnet2_utils.py:335-386): `idx` (B,M,K) given or the `nsample` nearest
    points of every centre; grouped coordinates relative to the centre; **xyz first**, features last (unlike QueryAndGroup).
    The reference's own neighbour search here is a call with the wrong arity (`knn(xyz, new_xyz, radius, nsample)` against
    `KNN.forward(k, unknown, known)`, l.361 vs l.80 -- it can only run with `idx` passed in); this module runs the search it
    evidently means: the `nsample` nearest points of `xyz` for every `new_xyz` (captra_knn)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor = None, idx: torch.Tensor = None,
                features: torch.Tensor = None) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,M,3) (default: xyz), idx (B,M,K) int32 or None, features (B,C,N) or None
        -> (B,3+C,M,K) (use_xyz) or (B,C,M,K)."""
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            _, idx = knn(self.nsample, new_xyz, xyz)
        idx = idx.detach().int()
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)           # (B,3,M,K)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
