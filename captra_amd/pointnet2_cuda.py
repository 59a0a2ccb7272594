"""Drop-in replacement for the reference's pybind11 module `pointnet2_cuda`.

Same ten function names and positional argument lists as
network/models/pointnet_lib/src/pointnet2_api.cpp:10-25, so that the reference's own
`pointnet_lib/pointnet2_utils.py` (`import pointnet2_cuda as pointnet2`, l.7) runs unmodified on
top of the HIP kernels once this module is importable under that name
(`captra_amd.install_as_pointnet2_cuda()` or put this directory on sys.path).

Ownership as in the reference: the caller allocates every tensor, the callee only writes
(SURVEY.md §8b).  Differences, all on the safe side: inputs are validated (device, dtype,
contiguity) and failures raise RuntimeError instead of calling exit(-1)
(e.g. ball_query_gpu.cu:62-66).
"""
from __future__ import annotations

import torch

from . import _lib as L


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise L.CaptraHipError(f"{name} must be a CUDA(HIP) tensor")  # CHECK_CUDA, ball_query.cpp:10
    if not t.is_contiguous():
        raise L.CaptraHipError(f"{name} must be contiguous")         # CHECK_CONTIGUOUS, ball_query.cpp:11
    if t.dtype != dtype:
        raise L.CaptraHipError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _need(t: torch.Tensor, numel: int, name: str) -> None:
    if t.numel() < numel:
        raise L.CaptraHipError(f"{name} has {t.numel()} elements, kernel needs {numel}")


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """ball_query.cpp:14-25. new_xyz (B,M,3), xyz (B,N,3) f32 -> idx (B,M,nsample) i32."""
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz"); _chk(idx, torch.int32, "idx")
    _need(new_xyz, b * m * 3, "new_xyz"); _need(xyz, b * n * 3, "xyz"); _need(idx, b * m * nsample, "idx")
    with torch.cuda.device(xyz.device):
        L.call("captra_ball_query", b, n, m, float(radius), nsample, L.ptr(new_xyz), L.ptr(xyz), L.ptr(idx))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """group_points.cpp:25-36. points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample)."""
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx"); _chk(out, torch.float32, "out")
    _need(points, b * c * n, "points"); _need(idx, b * npoints * nsample, "idx"); _need(out, b * c * npoints * nsample, "out")
    with torch.cuda.device(points.device):
        L.call("captra_group_points", b, c, n, npoints, nsample, L.ptr(points), L.ptr(idx), L.ptr(out))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """group_points.cpp:11-22."""
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx"); _chk(grad_points, torch.float32, "grad_points")
    _need(grad_out, b * c * npoints * nsample, "grad_out"); _need(idx, b * npoints * nsample, "idx"); _need(grad_points, b * c * n, "grad_points")
    with torch.cuda.device(grad_out.device):
        # scratch for the atomic-free path is the caller's (include/captra_hip.h): allocated here with torch, 0 bytes = atomics
        need = int(L.lib().captra_group_points_grad_ws_bytes(b, c, n, npoints, nsample))
        ws = torch.empty(need, dtype=torch.uint8, device=grad_out.device) if need else None
        L.call("captra_group_points_grad_ws", b, c, n, npoints, nsample, L.ptr(grad_out), L.ptr(idx), L.ptr(grad_points), L.ptr(ws), need)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """sampling.cpp:11-21. points (B,C,N), idx (B,npoints) -> out (B,C,npoints)."""
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx"); _chk(out, torch.float32, "out")
    _need(points, b * c * n, "points"); _need(idx, b * npoints, "idx"); _need(out, b * c * npoints, "out")
    with torch.cuda.device(points.device):
        L.call("captra_gather_points", b, c, n, npoints, L.ptr(points), L.ptr(idx), L.ptr(out))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    """sampling.cpp:24-35."""
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx"); _chk(grad_points, torch.float32, "grad_points")
    _need(grad_out, b * c * npoints, "grad_out"); _need(idx, b * npoints, "idx"); _need(grad_points, b * c * n, "grad_points")
    with torch.cuda.device(grad_out.device):
        # scratch for the atomic-free, bit-reproducible path is the caller's (as for group_points_grad): 0 bytes = atomics
        need = int(L.lib().captra_gather_points_grad_ws_bytes(b, c, n, npoints))
        ws = torch.empty(need, dtype=torch.uint8, device=grad_out.device) if need else None
        L.call("captra_gather_points_grad_ws", b, c, n, npoints, L.ptr(grad_out), L.ptr(idx), L.ptr(grad_points), L.ptr(ws), need)
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """sampling.cpp:38-49. points (B,N,3), temp (B,N) pre-filled 1e10 -> idx (B,M) i32."""
    _chk(points, torch.float32, "points"); _chk(temp, torch.float32, "temp"); _chk(idx, torch.int32, "idx")
    _need(points, b * n * 3, "points"); _need(temp, b * n, "temp"); _need(idx, b * m, "idx")
    with torch.cuda.device(points.device):
        L.call("captra_furthest_point_sampling", b, n, m, L.ptr(points), L.ptr(temp), L.ptr(idx))
    return 1


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    """interpolate.cpp:26-36."""
    _chk(unknown, torch.float32, "unknown"); _chk(known, torch.float32, "known")
    _chk(dist2, torch.float32, "dist2"); _chk(idx, torch.int32, "idx")
    _need(unknown, b * n * 3, "unknown"); _need(known, b * m * 3, "known"); _need(dist2, b * n * k, "dist2"); _need(idx, b * n * k, "idx")
    with torch.cuda.device(unknown.device):
        L.call("captra_knn", b, n, m, k, L.ptr(unknown), L.ptr(known), L.ptr(dist2), L.ptr(idx))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """interpolate.cpp:14-24. Writes SQUARED distances (the sqrt is applied by pointnet2_utils.py:134)."""
    _chk(unknown, torch.float32, "unknown"); _chk(known, torch.float32, "known")
    _chk(dist2, torch.float32, "dist2"); _chk(idx, torch.int32, "idx")
    _need(unknown, b * n * 3, "unknown"); _need(known, b * m * 3, "known"); _need(dist2, b * n * 3, "dist2"); _need(idx, b * n * 3, "idx")
    with torch.cuda.device(unknown.device):
        L.call("captra_three_nn", b, n, m, L.ptr(unknown), L.ptr(known), L.ptr(dist2), L.ptr(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """interpolate.cpp:39-53. points (B,C,M), idx/weight (B,N,3) -> out (B,C,N)."""
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    _chk(weight, torch.float32, "weight"); _chk(out, torch.float32, "out")
    _need(points, b * c * m, "points"); _need(idx, b * n * 3, "idx"); _need(weight, b * n * 3, "weight"); _need(out, b * c * n, "out")
    with torch.cuda.device(points.device):
        L.call("captra_three_interpolate", b, c, m, n, L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """interpolate.cpp:55-68."""
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    _chk(weight, torch.float32, "weight"); _chk(grad_points, torch.float32, "grad_points")
    _need(grad_out, b * c * n, "grad_out"); _need(idx, b * n * 3, "idx"); _need(weight, b * n * 3, "weight"); _need(grad_points, b * c * m, "grad_points")
    with torch.cuda.device(grad_out.device):
        need = int(L.lib().captra_three_interpolate_grad_ws_bytes(b, c, n, m))
        ws = torch.empty(need, dtype=torch.uint8, device=grad_out.device) if need else None
        L.call("captra_three_interpolate_grad_ws", b, c, n, m, L.ptr(grad_out), L.ptr(idx), L.ptr(weight), L.ptr(grad_points), L.ptr(ws), need)
