"""numpy front-end of oracle/captra_oracle.c (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Argument order and allocation rules follow the reference's Python op layer
(network/models/pointnet_lib/pointnet2_utils.py): outputs are allocated here, idx is int32,
everything else float32, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libcaptra_oracle.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
        try:
            _lib = C.CDLL(str(_LIB_PATH))
        except OSError:
            subprocess.run(["make", "-B", "-C", str(_HERE)], check=True, capture_output=True)
            _lib = C.CDLL(str(_LIB_PATH))
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def furthest_point_sample(xyz, npoint, temp=None):
    """xyz (B,N,3) -> idx (B,npoint) int32; start index 0, lowest index wins ties."""
    xyz = _f(xyz)
    B, N, _ = xyz.shape
    idx = np.zeros((B, npoint), np.int32)
    if temp is None:
        temp = np.full((B, N), 1e10, np.float32)
    lib().oracle_fps(C.c_int(B), C.c_int(N), C.c_int(npoint), _p(xyz), _p(temp), _p(idx))
    return idx


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    lib().oracle_ball_query(C.c_int(B), C.c_int(N), C.c_int(M), C.c_float(radius), C.c_int(nsample),
                            _p(new_xyz), _p(xyz), _p(idx))
    return idx


def grouping_operation(features, idx):
    features, idx = _f(features), _i(idx)
    B, Cc, N = features.shape
    _, M, K = idx.shape
    out = np.empty((B, Cc, M, K), np.float32)
    lib().oracle_group_points(C.c_int(B), C.c_int(Cc), C.c_int(N), C.c_int(M), C.c_int(K), _p(features), _p(idx), _p(out))
    return out


def grouping_operation_grad(grad_out, idx, N):
    grad_out, idx = _f(grad_out), _i(idx)
    B, Cc, M, K = grad_out.shape
    g = np.zeros((B, Cc, N), np.float32)
    lib().oracle_group_points_grad(C.c_int(B), C.c_int(Cc), C.c_int(N), C.c_int(M), C.c_int(K), _p(grad_out), _p(idx), _p(g))
    return g


def gather_operation(features, idx):
    features, idx = _f(features), _i(idx)
    B, Cc, N = features.shape
    M = idx.shape[1]
    out = np.empty((B, Cc, M), np.float32)
    lib().oracle_gather_points(C.c_int(B), C.c_int(Cc), C.c_int(N), C.c_int(M), _p(features), _p(idx), _p(out))
    return out


def gather_operation_grad(grad_out, idx, N):
    grad_out, idx = _f(grad_out), _i(idx)
    B, Cc, M = grad_out.shape
    g = np.zeros((B, Cc, N), np.float32)
    lib().oracle_gather_points_grad(C.c_int(B), C.c_int(Cc), C.c_int(N), C.c_int(M), _p(grad_out), _p(idx), _p(g))
    return g


def three_nn(unknown, known):
    """returns (dist2 SQUARED, idx) exactly as the kernel writes them."""
    unknown, known = _f(unknown), _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.empty((B, N, 3), np.float32)
    idx = np.empty((B, N, 3), np.int32)
    lib().oracle_three_nn(C.c_int(B), C.c_int(N), C.c_int(M), _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def knn(k, unknown, known):
    unknown, known = _f(unknown), _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.empty((B, N, k), np.float32)
    idx = np.empty((B, N, k), np.int32)
    rc = lib().oracle_knn(C.c_int(B), C.c_int(N), C.c_int(M), C.c_int(k), _p(unknown), _p(known), _p(d2), _p(idx))
    if rc != 0:
        raise ValueError("k must be in 1..200")
    return d2, idx


def three_interpolate(features, idx, weight):
    features, idx, weight = _f(features), _i(idx), _f(weight)
    B, Cc, M = features.shape
    N = idx.shape[1]
    out = np.empty((B, Cc, N), np.float32)
    lib().oracle_three_interpolate(C.c_int(B), C.c_int(Cc), C.c_int(M), C.c_int(N), _p(features), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, idx, weight = _f(grad_out), _i(idx), _f(weight)
    B, Cc, N = grad_out.shape
    g = np.zeros((B, Cc, M), np.float32)
    lib().oracle_three_interpolate_grad(C.c_int(B), C.c_int(Cc), C.c_int(N), C.c_int(M), _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def canonicalize(pts, mean, rot, trans, scale, P=1):
    """pts (B,3,N), mean (B,3), rot (B*P,3,3), trans (B*P,3), scale (B*P) -> (cn (B*P,3,N), n3 (B*P,N,3))."""
    pts, mean, rot, trans, scale = _f(pts), _f(mean).reshape(-1, 3), _f(rot).reshape(-1, 3, 3), _f(trans).reshape(-1, 3), _f(scale).reshape(-1)
    B, _, N = pts.shape
    cn = np.empty((B * P, 3, N), np.float32)
    n3 = np.empty((B * P, N, 3), np.float32)
    lib().oracle_canonicalize(C.c_int(B), C.c_int(P), C.c_int(N), _p(pts), _p(mean), _p(rot), _p(trans), _p(scale), _p(cn), _p(n3))
    return cn, n3


def pointwise_mlp(x, wt, bias, act=1):
    """x (B,cin,L), wt (cin,cout), bias (cout) -> (B,cout,L); exact fmaf chain, k ascending."""
    x, wt, bias = _f(x), _f(wt), _f(bias)
    B, cin = x.shape[:2]
    L = int(np.prod(x.shape[2:]))
    cout = wt.shape[1]
    y = np.empty((B, cout) + x.shape[2:], np.float32)
    lib().oracle_pointwise_mlp(C.c_int(B), C.c_int(cin), C.c_int(cout), C.c_longlong(L), _p(x), _p(wt), _p(bias), C.c_int(act), _p(y))
    return y


def sa_group(feat, xyz_cn, new_xyz, idx):
    """group + centre-subtract + concat [feat, xyz] -> (B, cfeat+3, M, K)."""
    xyz_cn, new_xyz, idx = _f(xyz_cn), _f(new_xyz), _i(idx)
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    cfeat = 0 if feat is None else feat.shape[1]
    feat = None if feat is None else _f(feat)
    x = np.empty((B, cfeat + 3, M, K), np.float32)
    lib().oracle_sa_group(C.c_int(B), C.c_int(N), C.c_int(M), C.c_int(K), C.c_int(cfeat), _p(feat), _p(xyz_cn), _p(new_xyz), _p(idx), _p(x))
    return x


def max_over_k(x, y=None, co_off=0):
    x = _f(x)
    B, Cc, M, K = x.shape
    if y is None:
        y = np.empty((B, Cc, M), np.float32)
    lib().oracle_max_over_k(C.c_int(B), C.c_int(Cc), C.c_int(M), C.c_int(K), _p(x), _p(y), C.c_int(y.shape[1]), C.c_int(co_off))
    return y


def fp_interpolate_concat(unknown, known, skip, feat_known):
    unknown, known, feat_known = _f(unknown), _f(known), _f(feat_known)
    B, N, _ = unknown.shape
    S = known.shape[1]
    c2 = feat_known.shape[1]
    c1 = 0 if skip is None else skip.shape[1]
    skip = None if skip is None else _f(skip)
    out = np.empty((B, c1 + c2, N), np.float32)
    lib().oracle_fp_interpolate_concat(C.c_int(B), C.c_int(N), C.c_int(S), C.c_int(c1), C.c_int(c2), _p(unknown), _p(known), _p(skip), _p(feat_known), _p(out))
    return out


def part_fit_st(labels, src, tgt, rot, sym, given_scale=None):
    """labels (B,N) int, src (B,P,3,N), tgt (B,3,N) or (B,P,3,N), rot (B,P,3,3)
    -> scale (B,P), trans (B,P,3), valid (B,P)."""
    labels, src, tgt, rot = _i(labels), _f(src), _f(tgt), _f(rot)
    B, P, _, N = src.shape
    per_part = 1 if tgt.ndim == 4 else 0
    gs = None if given_scale is None else _f(given_scale)
    scale = np.empty((B, P), np.float32)
    trans = np.empty((B, P, 3), np.float32)
    valid = np.empty((B, P), np.int32)
    lib().oracle_part_fit_st(C.c_int(B), C.c_int(P), C.c_int(N), C.c_int(1 if sym else 0), _p(labels), _p(src), _p(tgt),
                             C.c_int(per_part), _p(rot), _p(gs), _p(scale), _p(trans), _p(valid))
    return scale, trans, valid


def procrustes_rot3(src, tgt):
    """src, tgt (nb,N,3) -> rot (nb,3,3)."""
    src, tgt = _f(src), _f(tgt)
    nb, N, _ = src.shape
    rot = np.empty((nb, 3, 3), np.float32)
    lib().oracle_procrustes_rot3(C.c_int(nb), C.c_int(N), _p(src), _p(tgt), _p(rot))
    return rot
