"""Tensor-level wrappers of the fused tracking-path kernels (Section 2 of include/captra_hip.h).

Each function allocates its output with torch (caching allocator = plumbing), validates device /
contiguity, and launches on torch's current stream.  No CPU fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading

import torch

from . import _lib as L
from .fold import PackedLinear

ACT_NONE, ACT_RELU, ACT_SIGMOID_M05 = 0, 1, 2

# Algorithmic work accounting (bench.py's roofline object): per kernel family, the flops / bytes the
# launches of this process asked for.  Off by default; costs one dict update per launch when on.
WORK = {"on": False, "flops": {}, "bytes": {}}


def work_reset(on: bool = True) -> None:
    WORK["on"] = on
    WORK["flops"].clear()
    WORK["bytes"].clear()


def _work(name: str, flops: float = 0.0, nbytes: float = 0.0) -> None:
    if WORK["on"]:
        WORK["flops"][name] = WORK["flops"].get(name, 0.0) + flops
        WORK["bytes"][name] = WORK["bytes"].get(name, 0.0) + nbytes


def canonicalize(pts, mean, rot, trans, scale, num_parts: int = 1, want_cn=True, want_n3=True, want_planes=False):
    """pts (B,3,N), mean (B,3[,1]), rot (B*P,3,3), trans (B*P,3[,1]), scale (B*P) ->
    (out_cn (B*P,3,N), out_n3 (B*P,N,3)) = R^T((pts+mean)-t)/s  (networks.py:38-41, 184-187).
    want_planes: a third element, the clouds in the ball query's LDS plane order (B*P,3,pad256(N)) for the level-1 stream kernel."""
    B, _, N = pts.shape
    Q = B * num_parts
    mean = mean.reshape(B, 3).contiguous()
    rot = rot.reshape(Q, 3, 3).contiguous()
    trans = trans.reshape(Q, 3).contiguous()
    scale = scale.reshape(Q).contiguous()
    L.require_device(pts, mean, rot, trans, scale)
    out_cn = torch.empty(Q, 3, N, dtype=torch.float32, device=pts.device) if want_cn else None
    out_n3 = torch.empty(Q, N, 3, dtype=torch.float32, device=pts.device) if want_n3 else None
    if want_planes:
        planes = torch.empty(Q, 3, (N + 255) // 256 * 256, dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            L.call("captra_canonicalize_planes", B, num_parts, N, L.ptr(pts), L.ptr(mean), L.ptr(rot), L.ptr(trans), L.ptr(scale),
                   L.ptr(out_cn), L.ptr(out_n3), L.ptr(planes))
        return out_cn, out_n3, planes
    with torch.cuda.device(pts.device):
        L.call("captra_canonicalize", B, num_parts, N, L.ptr(pts), L.ptr(mean), L.ptr(rot), L.ptr(trans), L.ptr(scale),
               L.ptr(out_cn), L.ptr(out_n3))
    return out_cn, out_n3


SPLIT_K_MAX_TRAJECTORIES = int(os.environ.get("CAPTRA_SPLIT_K_TRAJ", "2"))   # 0 = every dense layer the k-ascending chain at every batch
SPLIT_K_POSITIONS = int(os.environ.get("CAPTRA_SPLIT_K_POSITIONS", "8192"))    # launches of at most this many positions (b * l) split k
# f32x6 mode, any number of trajectories: the layers that stay on the exact fp32 MFMA (SA3 / FP3 / FP2 / the SA2 pre-transform: 128 ..
# 512 points per cloud) are single dependent chains of cin / 2 MFMAs per wave, and the mode has no bit-exact contract, so they MAY split
# k.  Measured (tools/ab_round6.sh, 32 trajectories in two lanes, same box): 3.376 ms off, 3.378 at 2048 positions, 3.40-3.41 at 8192 /
# 16384 -- the other lane's kernels already fill the chip under those chains and the 32 x 32 split tiles re-read more.  Off.
X6_SPLIT_K_POSITIONS = int(os.environ.get("CAPTRA_X6_SPLIT_K_POSITIONS", "0"))
_split_k_on = False          # (kept for readers of the module attribute; the live flag is per thread: _split_k_active())


def _split_k_active() -> bool:
    return getattr(_TLS, "split_k_on", False)


def _split_k_limit() -> int:
    """Position limit (b * l) of the split-k form on this thread, 0 outside `split_k`."""
    return getattr(_TLS, "split_k_limit", 0) if _split_k_active() else 0


def split_k_rule(n_traj: int, allow_few: bool = True) -> int:
    """The position limit a track step of `n_traj` trajectories runs its dense layers under (0 = the k-ascending chain everywhere):
    SPLIT_K_POSITIONS for one or two trajectories (`allow_few`: not for the lanes of a larger exact-fp32 batch, which must repeat the
    eager step's bits), X6_SPLIT_K_POSITIONS for every f32x6 step."""
    if n_traj <= 0 or not exact_path():
        return 0
    if allow_few and n_traj <= SPLIT_K_MAX_TRAJECTORIES:
        return SPLIT_K_POSITIONS
    return X6_SPLIT_K_POSITIONS if mlp_dtype() == "f32x6" else 0


@contextlib.contextmanager
def split_k(on):
    """Dense layers launched inside with at most `on` positions (True: SPLIT_K_POSITIONS) split k over a workgroup's four waves
    (captra_launch_opts::splitk_positions of every call: a fixed summation order, 1e-5 relative from the bit-exact chain).  The
    track step of one or two trajectories runs under it -- its 128- / 512-point levels are single dependent MFMA chains on an idle
    chip otherwise -- and every f32x6 step (`split_k_rule`).  Re-entrant and per thread (the options are the host layer's, the C
    library keeps no state)."""
    prev = _split_k_active()
    limit = SPLIT_K_POSITIONS if on is True else int(on or 0)
    if limit > 0 and not prev:
        _TLS.split_k_on = True
        _TLS.split_k_limit = limit
        with L.launch_options(splitk_positions=limit):
            try:
                yield
            finally:
                _TLS.split_k_on = False
                _TLS.split_k_limit = 0
    else:
        yield


@contextlib.contextmanager
def centre_window(m0: int, mc: int):
    """The set-abstraction launches inside process centres [m0, m0 + mc) of every cloud only (captra_launch_opts::centre_m0 /
    centre_mc): the ball query and the small-input SA scales of the centres a streamed sampler (`fps_gather_part`) has picked so far."""
    with L.launch_options(centre_m0=int(m0), centre_mc=int(mc)):
        yield


def ball_query_multi(radii, nsamples, xyz_n3, new_xyz_n3, outs=None):
    """One scan of xyz for all radii -> [idx_r (B,M,K_r) int32] (`outs`: the lists to fill -- under `centre_window` a launch
    fills its window of them)."""
    L.require_device(xyz_n3, new_xyz_n3)
    B, N, _ = xyz_n3.shape
    M = new_xyz_n3.shape[1]
    nr = len(radii)
    if outs is None:
        outs = [torch.empty(B, M, int(k), dtype=torch.int32, device=xyz_n3.device) for k in nsamples]
    c_r = (C.c_float * nr)(*[float(r) for r in radii])
    c_k = (C.c_int * nr)(*[int(k) for k in nsamples])
    c_p = (C.c_void_p * nr)(*[o.data_ptr() for o in outs])
    with torch.cuda.device(xyz_n3.device):
        L.call("captra_ball_query_multi", B, N, M, nr, C.cast(c_r, C.c_void_p), C.cast(c_k, C.c_void_p),
               L.ptr(new_xyz_n3), L.ptr(xyz_n3), C.cast(c_p, C.c_void_p))
    # ball_query = 12N + 12M + 4MK bytes per cloud and radius
    # (`flops` carries the scan's N x M pair tests -- one scan serves every radius; the early exit makes it an upper bound)
    _work("ball_query", flops=float(B) * N * M, nbytes=B * sum(12 * N + 12 * M + 4 * M * int(k) for k in nsamples))
    return outs


def group_points_multi(points_list, idx_list, outs=None):
    """[grouping_operation(points_j (B,C_j,N), idx_j (B,M_j,K_j))] for jobs over clouds of the same size in ONE launch
    (captra_group_points_multi): every radius x every feature tensor of a set-abstraction level."""
    nj = len(points_list)
    assert nj == len(idx_list) and nj > 0
    B, _, N = points_list[0].shape
    if outs is None:
        outs = [torch.empty(B, p.shape[1], i.shape[1], i.shape[2], dtype=torch.float32, device=p.device) for p, i in zip(points_list, idx_list)]
    for p, i, o in zip(points_list, idx_list, outs):
        L.require_device(p, i, o)
        assert p.shape[0] == B and p.shape[2] == N and i.dtype == torch.int32 and tuple(o.shape) == (B, p.shape[1], i.shape[1], i.shape[2])
    arr = lambda ctype, vals: C.cast((ctype * nj)(*vals), C.c_void_p)      # noqa: E731
    with torch.cuda.device(points_list[0].device):
        L.call("captra_group_points_multi", B, N, nj, arr(C.c_int, [p.shape[1] for p in points_list]), arr(C.c_int, [i.shape[1] for i in idx_list]),
               arr(C.c_int, [i.shape[2] for i in idx_list]), arr(C.c_void_p, [p.data_ptr() for p in points_list]),
               arr(C.c_void_p, [i.data_ptr() for i in idx_list]), arr(C.c_void_p, [o.data_ptr() for o in outs]))
    return outs


def query_and_group(radius: float, nsample: int, xyz_n3, new_xyz_n3, features=None, use_xyz: bool = True, want_idx: bool = False):
    """The reference's QueryAndGroup module (pointnet_lib/pointnet2_utils.py:274-310) as ONE launch (captra_query_and_group): ball
    query, grouping of the coordinates, centre subtraction, grouping of the features and the concat.  xyz (B,N,3), new_xyz (B,M,3),
    features (B,C,N) or None -> (B, C + 3, M, K) (features first), [(B,M,K) int32 lists].  None when the shape is outside the
    kernel (nsample % 4, N > 8192): the caller runs the two ops."""
    L.require_device(xyz_n3, new_xyz_n3, features)
    B, N, _ = xyz_n3.shape
    M = new_xyz_n3.shape[1]
    C = 0 if features is None else features.shape[1]
    if nsample % 4 or N > 8192 or (features is None and not use_xyz):
        return None
    ct = C + (3 if (use_xyz or features is None) else 0)
    out = torch.empty(B, ct, M, nsample, dtype=torch.float32, device=xyz_n3.device)
    idx = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz_n3.device) if want_idx else None
    with torch.cuda.device(xyz_n3.device):
        L.call("captra_query_and_group", B, N, M, float(radius), nsample, C, 1 if use_xyz else 0, L.ptr(xyz_n3), L.ptr(new_xyz_n3),
               L.ptr(features), L.ptr(out), L.ptr(idx))
    return (out, idx) if want_idx else out


def seg_softmax_argmax(logits):
    """logits (B,S,N) -> (softmax over S (B,S,N), labels (B,N) int32 = first index of the largest logit): CoordinateNet's
    read-out (networks.py:50 + model.py:466) in one launch."""
    L.require_device(logits)
    B, S, N = logits.shape
    seg = torch.empty_like(logits)
    labels = torch.empty(B, N, dtype=torch.int32, device=logits.device)
    with torch.cuda.device(logits.device):
        L.call("captra_seg_softmax_argmax", B, S, N, L.ptr(logits), L.ptr(seg), L.ptr(labels))
    return seg, labels


def copy_multi(pairs) -> None:
    """[(src, dst)] device tensors of equal byte size (contiguous, 4-byte words), at most 16: all copies in ONE launch."""
    n = len(pairs)
    if n == 0:
        return
    for a, b in pairs:
        assert a.is_contiguous() and b.is_contiguous() and a.numel() * a.element_size() == b.numel() * b.element_size(), (a.shape, b.shape)
    arr = lambda ctype, vals: C.cast((ctype * n)(*vals), C.c_void_p)      # noqa: E731
    with torch.cuda.device(pairs[0][0].device):
        L.call("captra_copy_multi", n, arr(C.c_void_p, [a.data_ptr() for a, _ in pairs]), arr(C.c_void_p, [b.data_ptr() for _, b in pairs]),
               arr(C.c_longlong, [a.numel() * a.element_size() for a, _ in pairs]))


def row_max(x):
    """x (B,C,N) fp32 contiguous -> (B,C,1): max over the positions in one launch (captra_row_max)."""
    L.require_device(x)
    B, Cc, N = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(B, Cc, 1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_row_max", B * Cc, N, L.ptr(x), L.ptr(out))
    return out


def pack(wt_dense, bias_dense) -> PackedLinear:
    """Dense W^T (cin,cout) + bias (cout) -> the packed layout the kernels take."""
    return PackedLinear(wt_dense.float(), bias_dense.float())


# Arithmetic of the shared MLPs: "fp32" (exact, the metric's configuration) or "bf16" (bf16 MFMA operands / fp32 accumulation,
# BASELINE.json configs[2]; opt-in).  NOT process-global: a per-thread setting, normally entered by the model that owns it
# (EvalTrackModel(cfg['mlp_dtype']) wraps its step in `use_mlp_dtype`), so two models / two host threads never see each
# other's mode.
MLP_DTYPES = ("fp32", "bf16", "f32x6")
_TLS = threading.local()


def mlp_dtype() -> str:
    return getattr(_TLS, "mlp_dtype", "fp32")


def exact_path() -> bool:
    """The exact-fp32 kernels' routes: the default arithmetic, and the f32x6 mode for every layer without an f32x6 kernel."""
    return mlp_dtype() in ("fp32", "f32x6")


def set_mlp_dtype(dtype: str) -> None:
    """Default of the calling thread (tools / ad-hoc scripts); models use `use_mlp_dtype`."""
    if dtype not in MLP_DTYPES:
        raise ValueError(f"mlp dtype {dtype!r}: expected one of {MLP_DTYPES}")
    _TLS.mlp_dtype = dtype


@contextlib.contextmanager
def use_mlp_dtype(dtype):
    """`with use_mlp_dtype("bf16"): ...` -- the shared-MLP launches of this thread inside the block; None = leave as is."""
    if dtype is None:
        yield
        return
    prev = mlp_dtype()
    set_mlp_dtype(dtype)
    try:
        yield
    finally:
        _TLS.mlp_dtype = prev


def pointwise_mlp(x, lin: PackedLinear, act: int = ACT_RELU, out=None):
    """x (B,cin,*), packed layer -> (B,cout,*) = act(W x + b) (exact-fp32 MFMA; bf16 operands inside `use_mlp_dtype("bf16")`)."""
    wt, bias = lin.wt, lin.bias
    L.require_device(x, wt, bias)
    B, cin = x.shape[0], x.shape[1]
    cout = lin.cout
    assert lin.cin == cin, (lin.cin, x.shape)
    l = x.numel() // max(B * cin, 1)
    if out is None:
        out = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    if mlp_dtype() == "bf16":
        with torch.cuda.device(x.device):
            L.call("captra_pointwise_mlp_bf16", B, cin, cout, l, L.ptr(x), L.ptr(lin.bf16(0, cin)), L.ptr(bias), act, L.ptr(out))
        _work("pointwise_mlp", flops=2.0 * B * cin * cout * l, nbytes=4.0 * B * l * (cin + cout))
        return out
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp", B, cin, cout, l, L.ptr(x), L.ptr(wt), L.ptr(bias), act, L.ptr(out))
    _work("pointwise_mlp", flops=2.0 * B * cin * cout * l, nbytes=4.0 * B * l * (cin + cout))
    return out


def pointwise_mlp2(x, x2, lin: PackedLinear, act: int = ACT_RELU):
    """act(W [x; x2] + b) without building the concat (captra_pointwise_mlp2): x (B,c,l), x2 (B,c2,l) or (B,c2,1) (one vector per
    cloud, read for every position); exact fp32, bit-identical to pointwise_mlp on the concatenated tensor.  None when the shape is
    outside the kernel's range or the tensors lie too far apart for one buffer descriptor: the caller concatenates."""
    if not exact_path() or lin.cout <= 64 or x.dim() != 3 or x2.dim() != 3:
        return None
    L.require_device(x, x2)
    B, c, l = x.shape
    bcast = x2.shape[2] == 1 and l != 1
    assert lin.cin == c + x2.shape[1] and (bcast or x2.shape[2] == l), (x.shape, x2.shape, lin.cin)
    span = 4 * (x.numel() + x2.numel())
    if abs(x2.data_ptr() - x.data_ptr()) + span >= (1 << 30):
        return None
    out = torch.empty(B, lin.cout, l, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp2", B, lin.cin, c, lin.cout, l, L.ptr(x), L.ptr(x2), 1 if bcast else 0, L.ptr(lin.wt), L.ptr(lin.bias), act, L.ptr(out))
    _work("pointwise_mlp", flops=2.0 * B * lin.cin * lin.cout * l, nbytes=4.0 * B * l * (lin.cin + lin.cout))
    return out


X6_FP_HOIST = os.environ.get("CAPTRA_X6_FP_HOIST", "1") != "0"   # f32x6: a layer on [x; repeat(v)] as W1 x + (W2 v + b)


def pointwise_mlp_cloud_bias(x, v, lin: PackedLinear, act: int = ACT_RELU):
    """act(W [x; repeat(v)] + b) as act(W1 x + (W2 v + b)): x (B,c,l), v (B,c2,1) one vector per cloud.  The bracket is one small
    product per cloud (split-k: a single position is one dependent chain otherwise) and becomes the layer's bias per cloud
    (captra_pointwise_mlp_cb) -- c instead of c + c2 input channels at every position (FP3: 512 instead of 1536).  NOT the k-ascending
    chain over the concat, so only where the arithmetic has no bit-exact contract (the f32x6 mode); None outside its shapes."""
    # (not under the few-trajectory split-k rule: there the two-source layer itself splits k over a workgroup's waves)
    if not (X6_FP_HOIST and mlp_dtype() == "f32x6" and not _split_k_active() and x.dim() == 3 and v.dim() == 3 and v.shape[2] == 1 and lin.cout % 128 == 0
            and lin.cout > 64 and x.shape[1] >= 32 and lin.cin == x.shape[1] + v.shape[1]):
        return None
    L.require_device(x, v)
    B, c, l = x.shape
    with split_k(True):
        bias_bc = pointwise_mlp(v, lin.trailing_rows(c), ACT_NONE)             # (B,cout,1) = W2 v + b
    lead = lin.leading_rows(c)
    out = torch.empty(B, lin.cout, l, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp_cb", B, c, lin.cout, l, L.ptr(x), L.ptr(lead.wt), L.ptr(bias_bc), act, L.ptr(out))
    _work("pointwise_mlp", flops=2.0 * B * lin.cout * (c * l + v.shape[1]), nbytes=4.0 * B * l * (c + lin.cout))
    return out


def pm_channels(c: int) -> int:
    """Channel stride of a point-major bf16 tensor (include/captra_hip.h "bf16-NATIVE dense layers")."""
    return (c + 31) // 32 * 32


def pointwise_mlp_bf16pm_cloud_bias(x, lin: PackedLinear, l: int, bias_bc, out_pm: bool, act: int = ACT_NONE):
    """The bf16-native dense layer on a channel-major fp32 x (B,cin,l) with a bias per cloud, bias_bc (B,cout) fp32
    (captra_pointwise_mlp_bf16pm_cb): act(W x + bias_bc[b])."""
    B = x.shape[0]
    L.require_device(x, bias_bc)
    assert x.dtype == torch.float32 and x.shape[1] == lin.cin and x.numel() == B * lin.cin * l and lin.cout % 32 == 0
    assert bias_bc.dtype == torch.float32 and bias_bc.is_contiguous() and bias_bc.numel() == B * lin.cout
    y = (torch.empty(B, l, pm_channels(lin.cout), dtype=torch.bfloat16, device=x.device) if out_pm
         else torch.empty(B, lin.cout, l, dtype=torch.float32, device=x.device))
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp_bf16pm_cb", B, lin.cin, lin.cout, l, 0, L.ptr(x), L.ptr(lin.bf16_frag(False)), L.ptr(bias_bc), act,
               1 if out_pm else 0, L.ptr(y))
    _work("pointwise_mlp", flops=2.0 * B * lin.cin * lin.cout * l, nbytes=B * l * (4.0 * lin.cin + (2.0 if out_pm else 4.0) * lin.cout))
    return y


USE_STATS_EPILOGUE = os.environ.get("CAPTRA_STATS_EPILOGUE", "1") != "0"   # bf16 GroupNorm chains: the producing layer leaves the partial statistics of what it stored


def pointwise_mlp_bf16pm(x, lin: PackedLinear, l: int, in_pm: bool, out_pm: bool, ab=None, act: int = ACT_NONE, with_stats: bool = False):
    """One bf16-native dense layer (captra_pointwise_mlp_bf16pm).  x: (B,cin,l) fp32, or (B,l,ceil32(cin)) bf16 slot order when
    in_pm; result (B,cout,l) fp32, or (B,l,ceil32(cout)) bf16 slot order when out_pm.  ab (B,cin,2): GroupNorm coefficients of
    the producer, applied as relu(a x + b) on load.  with_stats (out_pm): -> (y, stats (B,T,cout,2), TILE-major), the partial (sum, sum of
    squares) of the STORED values per chunk of 64 positions, written by the layer's own epilogue (captra_pointwise_mlp_bf16pm_stats)."""
    B = x.shape[0]
    if with_stats:
        assert out_pm
        L.require_device(x, ab)
        y = torch.empty(B, l, pm_channels(lin.cout), dtype=torch.bfloat16, device=x.device)
        stats = torch.empty(B, L.lib().captra_dense_bf16_stats_tiles(l), lin.cout, 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.call("captra_pointwise_mlp_bf16pm_stats", B, lin.cin, lin.cout, l, 1 if in_pm else 0, L.ptr(x), L.ptr(lin.bf16_frag(in_pm)),
                   L.ptr(lin.bias), L.ptr(ab), act, L.ptr(y), L.ptr(stats))
        _work("pointwise_mlp", flops=2.0 * B * lin.cin * lin.cout * l, nbytes=B * l * ((2.0 if in_pm else 4.0) * lin.cin + 2.0 * lin.cout))
        return y, stats
    L.require_device(x, ab)
    if in_pm:
        assert x.dtype == torch.bfloat16 and tuple(x.shape) == (B, l, pm_channels(lin.cin)), (x.shape, lin.cin, l)
    else:
        assert x.dtype == torch.float32 and x.shape[1] == lin.cin and x.numel() == B * lin.cin * l, (x.shape, lin.cin, l)
    y = (torch.empty(B, l, pm_channels(lin.cout), dtype=torch.bfloat16, device=x.device) if out_pm
         else torch.empty(B, lin.cout, l, dtype=torch.float32, device=x.device))
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp_bf16pm", B, lin.cin, lin.cout, l, 1 if in_pm else 0, L.ptr(x), L.ptr(lin.bf16_frag(in_pm)),
               L.ptr(lin.bias), L.ptr(ab), act, 1 if out_pm else 0, L.ptr(y))
    _work("pointwise_mlp", flops=2.0 * B * lin.cin * lin.cout * l, nbytes=B * l * ((2.0 if in_pm else 4.0) * lin.cin + (2.0 if out_pm else 4.0) * lin.cout))
    return y


USE_TILE_BF16 = os.environ.get("CAPTRA_TILE_BF16", "1") != "0"   # point-major bf16 layers through the LDS-tiled kernel (csrc/tile_bf16.hip)


def dense_bf16_tile_supported(x, lin: PackedLinear) -> bool:
    """Point-major bf16 in / out layers the LDS-tiled kernel is instantiated for (captra_dense_bf16_tile)."""
    return (USE_TILE_BF16 and x.dtype == torch.bfloat16 and x.dim() == 3 and lin.cout >= 64
            and x.shape[1] * x.shape[2] * 2 < (1 << 31))


OUT_PM, OUT_CM, OUT_MAX = 0, 1, 2      # captra_dense_bf16_tile_ex out_mode


def dense_bf16_tile(x, lin: PackedLinear, ab=None, act: int = ACT_NONE, with_stats: bool = False, bias_bc=None, x2=None, l=None,
                    out_mode: int = OUT_PM):
    """One dense layer through the LDS-tiled kernel (captra_dense_bf16_tile_ex).  x: (B,l,ceil32(cin)) bf16 slot order, or -- fp32
    -- (B,c,l) channel-major, optionally with x2 (B,cin - c,l) holding the remaining input channels (the concat is never built).
    out_mode OUT_PM: y (B,l,ceil32(cout)) bf16 slot order [, stats (B,T,cout,2) tile-major: partial sums of the fp32 outputs per
    chunk of 128 positions]; OUT_CM: (B,cout,l) fp32; OUT_MAX: (B,cout,1) fp32, the max over the l <= 128 positions.  ab (B,cin,2):
    the producer's GroupNorm applied as relu(a x + b) while the operand is staged.  bias_bc (B,cout): a bias per cloud."""
    L.require_device(x, ab, bias_bc, x2)
    in_cm = x.dtype == torch.float32
    B = x.shape[0]
    if in_cm:
        csplit = x.shape[1]
        l = x.numel() // max(B * csplit, 1) if l is None else l
        assert csplit + (0 if x2 is None else x2.shape[1]) == lin.cin and (x2 is None or x2.dtype == torch.float32), (x.shape, lin.cin)
    else:
        l, csplit = x.shape[1], lin.cin
        assert x.dtype == torch.bfloat16 and x.shape[2] == pm_channels(lin.cin), (x.shape, lin.cin)
    if out_mode == OUT_PM:
        y = torch.empty(B, l, pm_channels(lin.cout), dtype=torch.bfloat16, device=x.device)
    elif out_mode == OUT_CM:
        y = torch.empty(B, lin.cout, l, dtype=torch.float32, device=x.device)
    else:
        y = torch.empty(B, lin.cout, 1, dtype=torch.float32, device=x.device)
    stats = (torch.empty(B, L.lib().captra_dense_bf16_tile_stats_tiles(l), lin.cout, 2, dtype=torch.float32, device=x.device)
             if with_stats else None)
    if bias_bc is not None:
        assert bias_bc.dtype == torch.float32 and bias_bc.numel() == B * lin.cout and lin.cout % 32 == 0
    with torch.cuda.device(x.device):
        L.call("captra_dense_bf16_tile_ex", B, lin.cin, lin.cout, l, 1 if in_cm else 0, L.ptr(x), L.ptr(x2), csplit, L.ptr(lin.bf16_frag(True)),
               L.ptr(lin.bias if bias_bc is None else bias_bc), 0 if bias_bc is None else lin.cout, L.ptr(ab), act, out_mode, L.ptr(y), L.ptr(stats))
    _work("pointwise_mlp", flops=2.0 * B * lin.cin * lin.cout * l,
          nbytes=B * l * ((4.0 if in_cm else 2.0) * lin.cin + (2.0 if out_mode == OUT_PM else 4.0 if out_mode == OUT_CM else 0.0) * lin.cout))
    return (y, stats) if with_stats else y


def gemv_bf16(v, lin: PackedLinear):
    """One vector per cloud through a layer: v (B,cin[,1]) fp32 -> (B,cout) fp32 = b + sum_k bf16(w) bf16(v) (captra_gemv_bf16)."""
    L.require_device(v)
    B = v.shape[0]
    assert v.dtype == torch.float32 and v.numel() == B * lin.cin
    y = torch.empty(B, lin.cout, dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        L.call("captra_gemv_bf16", B, lin.cin, lin.cout, L.ptr(v), L.ptr(lin.wt), L.ptr(lin.bias), L.ptr(y))
    return y


def chain_tile_bf16_supported(c0: int, l: int, layers, pool: bool = False) -> bool:
    """A run of dense layers on a channel-major fp32 input, every layer through the LDS-tiled kernel."""
    return (USE_TILE_BF16 and mlp_dtype() == "bf16" and l % 4 == 0 and l >= 4 and layers[0].cout >= 32
            and all(lin.cout >= 64 for lin in layers[1:]) and (not pool or l <= 128))


def mlp_chain_bf16_tile(x, layers, acts, x2=None, out_pm: bool = False, pool: bool = False, bias_bc=None):
    """x (B,c,l) fp32 [+ x2 (B,c',l): the input is their channel concat, never built] through `layers`, hidden activations bf16
    point-major in HBM; the result as (B,c_n,l) fp32, the point-major bf16 tensor (out_pm), or (B,c_n,1) = its max over the
    positions (pool).  bias_bc (B,c_1): a per-cloud bias for the first layer."""
    B = x.shape[0]
    l = x.numel() // max(B * x.shape[1], 1)
    y = x.contiguous()
    for i, (lin, act) in enumerate(zip(layers, acts)):
        last = i == len(layers) - 1
        mode = OUT_PM if (not last or out_pm) else (OUT_MAX if pool else OUT_CM)
        y = dense_bf16_tile(y, lin, act=act, x2=x2 if i == 0 else None, l=l, out_mode=mode, bias_bc=bias_bc if i == 0 else None)
    return y


USE_NECK_CHAIN = os.environ.get("CAPTRA_NECK_CHAIN", "1") != "0"   # a neck module's layers in ONE launch (csrc/neck_bf16.hip); 0 = layer by layer (A/B, tests)


def neck_chain_supported(c0: int, l: int, layers, csplit: int, kind: int) -> bool:
    """The launcher's own conditions (captra_neck_chain_bf16): bf16 mode, two or three layers, hidden widths 128-multiples up to
    256 then 512, every width a multiple of 32, whole position quads."""
    n = len(layers)
    if not (USE_NECK_CHAIN and USE_TILE_BF16 and mlp_dtype() == "bf16" and n in (2, 3) and l % 4 == 0 and l >= 4 and layers[0].cin == c0):
        return False
    w = [lin.cout for lin in layers]
    if any(c % 32 for c in w) or w[0] > 256 or w[0] % 128 or (n == 3 and (w[1] > 512 or w[1] % 128)) or not 0 <= csplit <= c0:
        return False
    mt = [(c // 32 + 7) // 8 for c in w]
    return (mt == [1, 2, 4] and kind == 0) or (mt == [1, 1] and kind in (1, 2))


def neck_chain(kind: int, x, layers, x2=None, nn=None, v=None, v_rows=None, act_last: int = ACT_RELU):
    """A module of the backbone's neck in one launch (captra_neck_chain_bf16).  kind 0 (SA3): x (B,c,L), x2 (B,c',L) -> (B,c_n,1) = relu(max
    over the positions); kind 1 (FP3): x (B,c0,L), v (B,cv[,1]) the cloud's pooled vector, v_rows the PackedLinear of the first layer's
    rows that multiply v -> (B,c_n,L); kind 2 (FP2): x (B,c,L) the skip features, x2 (B,c',S) known features, nn = (idx, weight) (B,L,3)
    -> (B,c_n,L).  Bit-identical to the layer-by-layer route."""
    L.require_device(x, x2, v)
    B, csplit, l = x.shape[0], x.shape[1], x.shape[2]
    n = len(layers)
    chans = (C.c_int * (n + 1))(layers[0].cin, *[lin.cout for lin in layers])
    P = C.c_void_p
    imgs = [lin.bf16_frag(True) for lin in layers]
    wp = (P * n)(*[t.data_ptr() for t in imgs])
    bp = (P * n)(*[lin.bias.data_ptr() for lin in layers])
    cout = layers[-1].cout
    y = torch.empty((B, cout, 1) if kind == 0 else (B, cout, l), dtype=torch.float32, device=x.device)
    idx = wgt = None
    s_known = 0
    if kind == 2:
        idx, wgt = nn
        L.require_device(idx, wgt)
        s_known = x2.shape[2]
    gw = None
    if v_rows is not None:
        # the rows that multiply the cloud's vector, rounded once (RNE: what the gemv kernels do to the fp32 rows on the fly)
        if "gemv16" not in v_rows._bf16:
            v_rows._bf16["gemv16"] = v_rows.wt2d[:v_rows.cin].to(torch.bfloat16).contiguous()
        gw = v_rows._bf16["gemv16"]
    with torch.cuda.device(x.device):
        L.call("captra_neck_chain_bf16", kind, B, l, n, chans, L.ptr(x), L.ptr(x2), csplit, wp, bp, L.ptr(idx), L.ptr(wgt), s_known,
               L.ptr(v), L.ptr(gw), 0 if v is None else v.shape[1], act_last, L.ptr(y))
    flops = 2.0 * B * l * sum(lin.cin * lin.cout for lin in layers)
    _work("neck_chain", flops=flops, nbytes=4.0 * B * l * (layers[0].cin + cout))
    return y


def head12_bf16_supported(x, lin1: PackedLinear, lin2: PackedLinear) -> bool:
    return (USE_TILE_BF16 and x.dtype == torch.bfloat16 and x.dim() == 3 and lin1.cin <= 128 and lin1.cout == 512 and lin2.cin == 512
            and lin2.cout == 512 and x.shape[1] * 512 * 2 < (1 << 31))


def head12_bf16_stats(x, lin1: PackedLinear):
    """Statistics pass of a head's first layer: x (B,l,ceil32(cin)) bf16 -> partial sums (B,T,512,2) of y1 = W1 x + b1 (never stored)."""
    L.require_device(x)
    B, l, cp = x.shape
    assert cp == pm_channels(lin1.cin)
    stats = torch.empty(B, L.lib().captra_dense_bf16_tile_stats_tiles(l), 512, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_head12_bf16", B, lin1.cin, l, L.ptr(x), L.ptr(lin1.bf16_frag(True)), L.ptr(lin1.bias), None, None, None, None, L.ptr(stats))
    _work("pointwise_mlp", flops=2.0 * B * lin1.cin * 512 * l, nbytes=2.0 * B * l * lin1.cin)
    return stats


def head12_bf16(x, lin1: PackedLinear, ab1, lin2: PackedLinear):
    """Layers 1 + 2 of a Conv -> GroupNorm -> ReLU head in one launch: -> (y2 (B,l,512) bf16 raw, y2's partial statistics)."""
    L.require_device(x, ab1)
    B, l, cp = x.shape
    y2 = torch.empty(B, l, 512, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(B, L.lib().captra_dense_bf16_tile_stats_tiles(l), 512, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_head12_bf16", B, lin1.cin, l, L.ptr(x), L.ptr(lin1.bf16_frag(True)), L.ptr(lin1.bias), L.ptr(ab1),
               L.ptr(lin2.bf16_frag(True)), L.ptr(lin2.bias), L.ptr(y2), L.ptr(stats))
    _work("pointwise_mlp", flops=2.0 * B * l * (lin1.cin * 512 + 512 * 512), nbytes=2.0 * B * l * (lin1.cin + 512))
    return y2, stats


def gn_stats_bf16pm(x, c: int):
    """x (B,l,ceil32(c)) bf16 slot order -> partial statistics (B,c,T,2) of the stored values (captra_gn_stats_bf16pm)."""
    L.require_device(x)
    B, l, cp = x.shape
    assert x.dtype == torch.bfloat16 and cp == pm_channels(c)
    t = L.lib().captra_gn_stats_bf16pm_tiles(l)
    stats = torch.empty(B, c, t, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_gn_stats_bf16pm", B, c, l, L.ptr(x), L.ptr(stats))
    _work("gn_stats", nbytes=2.0 * B * l * cp)
    return stats


def mlp_chain_bf16(x, layers, acts, out_pm: bool = False):
    """A run of dense layers in the bf16 mode with the hidden activations bf16 point-major in HBM: x (B,c0,*) fp32 ->
    acts[-1](W_n ... relu(W_1 x + b_1) ...) as (B,c_n,*) fp32, or the (B,l,ceil32(c_n)) bf16 slot-order tensor when out_pm."""
    B = x.shape[0]
    l = x.numel() // max(B * x.shape[1], 1)
    if chain_tile_bf16_supported(x.shape[1], l, layers):
        y = mlp_chain_bf16_tile(x.reshape(B, x.shape[1], l), layers, acts, out_pm=out_pm)
        return y if out_pm else y.view((B, layers[-1].cout) + tuple(x.shape[2:]))
    y, in_pm = x.contiguous(), False
    for i, (lin, act) in enumerate(zip(layers, acts)):
        last = i == len(layers) - 1
        y = pointwise_mlp_bf16pm(y, lin, l, in_pm=in_pm, out_pm=out_pm or not last, act=act)
        in_pm = True
    return y if out_pm else y.view((B, layers[-1].cout) + tuple(x.shape[2:]))


class PMTensor:
    """A bf16 point-major activation tensor handed from one fused module to the next in the bf16 mode: `data` (B,L,ceil32(C))
    bf16 in slot order (include/captra_hip.h), `channels` = C."""
    __slots__ = ("data", "channels")

    def __init__(self, data, channels: int):
        self.data, self.channels = data, channels


USE_CHAIN_BF16 = True     # FP1 + conv1 (+ CoordinateNet's heads) register-resident in one launch (captra_mlp_chain_bf16)


def chain_bf16_supported(x, layers, heads=None) -> bool:
    # (captra_mlp_chain_bf16 instantiates 8 and 9 input k-steps: 113 .. 144 input channels; anything else returns -2)
    if not (USE_CHAIN_BF16 and mlp_dtype() == "bf16" and len(layers) == 3 and x.dim() == 3 and (x.shape[1] + 15) // 16 in (8, 9)):
        return False
    if not all(lin.cout == 128 for lin in layers) or layers[1].cin != 128 or layers[2].cin != 128 or x.shape[1] * x.shape[2] * 4 >= (1 << 31):
        return False
    if heads is not None:
        seg, hid, out = heads
        return seg.cin == 128 and hid.cin == 128 and hid.cout == 128 and out.cin == 128 and seg.cout <= 32 and out.cout <= 32
    return True


def _chain_bf16_image(layers, heads):
    """The chain's weight image (fragments of every layer back to back, then 32 bias floats per row tile), built once and cached
    with the first layer."""
    alls = list(layers) + (list(heads) if heads is not None else [])
    key = ("chain_img",) + tuple(id(l) for l in alls)
    cache = layers[0]._bf16
    if key not in cache:
        parts = [lin.bf16_frag(i > 0) for i, lin in enumerate(alls)]
        biases = [lin.bias[:pm_channels(lin.cout)].contiguous().view(torch.uint8) for lin in alls]
        img = torch.cat(parts + biases).contiguous()
        assert img.numel() == L.lib().captra_chain_bf16_image_bytes(layers[0].cin, 1 if heads is not None else 0), img.numel()
        cache[key] = (img, alls)
    return cache[key][0]


def mlp_chain_bf16_fused(x, layers, heads=None):
    """x (B,c0,L) fp32 through three 128-wide Conv+BN+ReLU layers in ONE launch.  heads None -> the feature map as a PMTensor;
    heads = (seg, hidden, out) packed layers -> (seg logits (B,S,L), sigmoid(nocs) - 0.5 (B,3P,L)) fp32."""
    L.require_device(x)
    B, c0, l = x.shape
    img = _chain_bf16_image(layers, heads)
    if heads is None:
        feat = torch.empty(B, l, 128, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            L.call("captra_mlp_chain_bf16", B, c0, l, 0, 0, 0, L.ptr(x), L.ptr(img), L.ptr(feat), None, None)
        _work("mlp_chain3", flops=2.0 * B * l * (c0 * 128 + 2 * 128 * 128), nbytes=B * l * (4.0 * c0 + 2.0 * 128))
        return PMTensor(feat, 128)
    seg = torch.empty(B, heads[0].cout, l, dtype=torch.float32, device=x.device)
    nocs = torch.empty(B, heads[2].cout, l, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_mlp_chain_bf16", B, c0, l, 1, heads[0].cout, heads[2].cout, L.ptr(x), L.ptr(img), None, L.ptr(seg), L.ptr(nocs))
    _work("coord_tail", flops=2.0 * B * l * (c0 * 128 + 3 * 128 * 128 + 128 * (heads[0].cout + heads[2].cout)),
          nbytes=4.0 * B * l * (c0 + heads[0].cout + heads[2].cout))
    return seg, nocs


def gn_chain_bf16_supported(x, couts) -> bool:
    """Conv -> GroupNorm -> ReLU chains in the bf16 mode: every normalised width must tile the statistics kernel."""
    return mlp_dtype() == "bf16" and x.dim() == 3 and all(256 % (pm_channels(c) // 8) == 0 for c in couts)


def fps_gather(xyz_n3, m: int, n_per_cloud=None):
    """xyz (B,N,3) -> (idx (B,m) int32, new_xyz (B,m,3), new_xyz (B,3,m)): sampling and the gather of the sampled
    coordinates in one launch; None when the cloud is too large for that kernel (caller samples and gathers separately).
    n_per_cloud (B,) int32 device tensor: ragged batch, cloud i samples from its first n_per_cloud[i] points."""
    L.require_device(xyz_n3)
    B, N, _ = xyz_n3.shape
    if N > 16 * 64 * 32:              # beyond 32 points per lane x 16 waves the cloud no longer fits the register file
        return None
    idx = torch.empty(B, m, dtype=torch.int32, device=xyz_n3.device)
    n3 = torch.empty(B, m, 3, dtype=torch.float32, device=xyz_n3.device)
    cn = torch.empty(B, 3, m, dtype=torch.float32, device=xyz_n3.device)
    with torch.cuda.device(xyz_n3.device):
        if n_per_cloud is None:
            L.call("captra_fps_gather", B, N, m, L.ptr(xyz_n3), L.ptr(idx), L.ptr(n3), L.ptr(cn))
        else:
            assert n_per_cloud.dtype == torch.int32 and n_per_cloud.numel() == B and n_per_cloud.is_contiguous()
            L.require_device(n_per_cloud)
            L.call("captra_fps_gather_ragged", B, N, L.ptr(n_per_cloud), m, L.ptr(xyz_n3), L.ptr(idx), L.ptr(n3), L.ptr(cn))
    return idx, n3, cn


def fps_gather_parts(xyz_n3, m: int):
    """Buffers of a STREAMED sampling (captra_fps_gather_part): (idx (B,m), new_xyz (B,m,3), new_xyz (B,3,m), state (B,N)); None
    when the cloud is outside the register-resident kernel."""
    B, N, _ = xyz_n3.shape
    if N >= 8192 or N > 16 * 64 * 32:
        return None
    dev = xyz_n3.device
    return (torch.empty(B, m, dtype=torch.int32, device=dev), torch.empty(B, m, 3, dtype=torch.float32, device=dev),
            torch.empty(B, 3, m, dtype=torch.float32, device=dev), torch.empty(B, N, dtype=torch.float32, device=dev))


def fps_gather_part(xyz_n3, m: int, j0: int, j1: int, bufs):
    """Picks [j0, j1) of the m into `bufs` (fps_gather_parts); parts in order on one stream == fps_gather, bit for bit."""
    idx, n3, cn, state = bufs
    L.require_device(xyz_n3, idx, n3, cn, state)
    B, N, _ = xyz_n3.shape
    with torch.cuda.device(xyz_n3.device):
        L.call("captra_fps_gather_part", B, N, m, j0, j1, L.ptr(xyz_n3), L.ptr(state), L.ptr(idx), L.ptr(n3), L.ptr(cn))


# CAPTRA's first set-abstraction level (configs/pointnet_config/pointnet2_*.yml sa1): what the level-1 stream kernel is built for
_L1_STREAM_WIDTHS = ((32, 32, 64), (64, 64, 128), (64, 96, 128))
_L1_STREAM_K = (32, 64, 128)
USE_L1_STREAM = os.environ.get("CAPTRA_L1_STREAM", "1") != "0"
# clouds per launch up to which the stream kernel pays (same-box A/B, bf16 step: 16 per lane 1.277 -> 1.23 ms, 32 in one lane 1.28 ->
# 1.265; two lanes of 32 lose 2 %: the consumers' backlog behind the sampler grows with the batch while the sampler's time does not)
L1_STREAM_MAX_CLOUDS = int(os.environ.get("CAPTRA_L1_STREAM_MAX", "32"))


def sa1_stream_supported(n: int, sa_modules, cfeats) -> bool:
    """`sa_modules`: the PointNetSetAbstractionMsg modules (one or two) whose level runs in the stream kernel, `cfeats` their
    feature-channel counts.  bf16 mode, CAPTRA's SA1 shapes, clouds of <= 4096 points."""
    if not (USE_L1_STREAM and mlp_dtype() == "bf16" and 1 <= len(sa_modules) <= 2 and n <= 4096):
        return False
    first = sa_modules[0]
    for mod, cf in zip(sa_modules, cfeats):
        if mod.training or mod.knn or cf not in (0, 3) or tuple(mod.nsample_list) != _L1_STREAM_K or mod.npoint > 512 or mod.npoint % 32:
            return False
        if list(mod.radius_list) != sorted(mod.radius_list):
            return False        # the kernel's scan rejects against the LAST radius and fills the lists in list order: radii ascending
        if mod.npoint != first.npoint or tuple(mod.radius_list) != tuple(first.radius_list):
            return False
        folded = mod._folded
        if folded is None or tuple(tuple(l.cout for l in layers) for layers in folded) != _L1_STREAM_WIDTHS:
            return False
    return True


def bq_planes(xyz_n3):
    """(B,N,3) -> (B,3,pad256(N)): the clouds in the ball query's LDS plane order (captra_bq_planes)."""
    L.require_device(xyz_n3)
    B, N, _ = xyz_n3.shape
    out = torch.empty(B, 3, (N + 255) // 256 * 256, dtype=torch.float32, device=xyz_n3.device)
    with torch.cuda.device(xyz_n3.device):
        L.call("captra_bq_planes", B, N, L.ptr(xyz_n3), L.ptr(out))
    return out


def sa1_stream_bf16(xyz_n3, xyz_cn, sa_modules, feats, planes=None, m2: int = 0):
    """Level 1 of the backbones in `sa_modules` on the SAME cloud in one launch (captra_sa1_stream_bf16): sampling, the three ball
    queries and every network's pooled SA1 features.  feats[i]: (B,cf,N) or None.  -> (fps_idx, new_xyz (B,M,3), new_xyz (B,3,M),
    [idx (B,M,K)] * 3, [out (B,320,M)] per network, scratch[, level 2]).  scratch[-16:].view(int32)[1] != 0 after completion = a consumer
    gave up.  m2 > 0: the second level's sampling too -- (idx2 (B,m2), new_xyz2 (B,m2,3), new_xyz2 (B,3,m2)) as the last element."""
    L.require_device(xyz_n3, xyz_cn, *[f for f in feats if f is not None])
    B, N, _ = xyz_n3.shape
    first = sa_modules[0]
    M = first.npoint
    dev = xyz_n3.device
    idx = torch.empty(B, M, dtype=torch.int32, device=dev)
    n3 = torch.empty(B, M, 3, dtype=torch.float32, device=dev)
    cn = torch.empty(B, 3, M, dtype=torch.float32, device=dev)
    lists = [torch.empty(B, M, k, dtype=torch.int32, device=dev) for k in _L1_STREAM_K]
    outs = [torch.empty(B, 320, M, dtype=torch.float32, device=dev) for _ in sa_modules]
    nbytes = L.lib().captra_sa1_stream_scratch_bytes(B, M)
    scratch = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    cfs = [0 if f is None else f.shape[1] for f in feats]
    imgs = [[sa_bf16_image(layers, cf, False) for layers in mod._folded] for mod, cf in zip(sa_modules, cfs)]
    lvl2 = None
    if m2 > 0:
        lvl2 = (torch.empty(B, m2, dtype=torch.int32, device=dev), torch.empty(B, m2, 3, dtype=torch.float32, device=dev),
                torch.empty(B, 3, m2, dtype=torch.float32, device=dev))
    P = C.c_void_p
    lp = (P * 3)(*[t.data_ptr() for t in lists])
    ia = (P * 3)(*[t.data_ptr() for t in imgs[0]])
    ib = (P * 3)(*[t.data_ptr() for t in imgs[1]]) if len(imgs) > 1 else (P * 3)()
    rad = (C.c_float * 3)(*[float(r) for r in first.radius_list])
    with torch.cuda.device(dev):
        L.call("captra_sa1_stream_bf16", B, N, M, L.ptr(xyz_n3), L.ptr(xyz_cn), L.ptr(planes), rad, L.ptr(idx), L.ptr(n3), L.ptr(cn), lp,
               cfs[0], L.ptr(feats[0]), ia, L.ptr(outs[0]), cfs[1] if len(cfs) > 1 else -1, L.ptr(feats[1]) if len(feats) > 1 else None,
               ib, L.ptr(outs[1]) if len(outs) > 1 else None, m2, L.ptr(lvl2[0]) if lvl2 else None, L.ptr(lvl2[1]) if lvl2 else None,
               L.ptr(lvl2[2]) if lvl2 else None, L.ptr(scratch))
    for mod, cf in zip(sa_modules, cfs):
        for (l1, l2, l3), k in zip(mod._folded, _L1_STREAM_K):
            _work("sa_scale_fused", flops=2.0 * B * M * k * ((cf + 3) * l1.cout + l1.cout * l2.cout + l2.cout * l3.cout),
                  nbytes=4.0 * B * (cf * N + 3 * N + M * k + 3 * M + l3.cout * M))
    if lvl2 is not None:
        return idx, n3, cn, lists, outs, scratch, lvl2
    return idx, n3, cn, lists, outs, scratch


def sa1_stream_spans(scratch):
    """(sampler span, whole span) in us of the launch that used `scratch`, from the kernel's own 100 MHz stamps (synchronises)."""
    w = scratch.view(torch.int32)[-16:].cpu().numpy().astype("uint32")
    start = (~w[2]) & 0xFFFFFFFF
    return float((int(w[3]) - int(start)) & 0xFFFFFFFF) / 100.0, float((int(w[4]) - int(start)) & 0xFFFFFFFF) / 100.0


def sa1_stream_gave_up(scratch) -> bool:
    """True when a consumer of the stream kernel's launch that used `scratch` gave up waiting (synchronises)."""
    return bool(scratch.view(torch.int32)[-15].item())


USE_SA_PRE = True        # SA scales with many feature channels: first layer's feature part once per source point
_SA_PRE_SHAPES = {(320, 128, 128, 256), (320, 128, 196, 256)}   # csrc/sa_fused.hip SWP_CASE list


def sa_scale_pre_supported(cfeat, layers, k) -> bool:
    return (USE_SA_PRE and USE_SA_FUSED and len(layers) == 3 and k % 32 == 0 and 128 % k == 0
            and (cfeat, layers[0].cout, layers[1].cout, layers[2].cout) in _SA_PRE_SHAPES)


_SA_BF16_SHAPES = {(0, 32, 32, 64, 32), (0, 64, 64, 128, 64), (0, 64, 96, 128, 128), (3, 32, 32, 64, 32), (3, 64, 64, 128, 64),
                   (3, 64, 96, 128, 128), (320, 128, 128, 256, 64), (320, 128, 196, 256, 128)}      # csrc/sa_bf16.hip SB_CASE list


# small-input scales whose kernels (sa_wave_lds_kernel / sa_bf16_kernel) process a centre window: (cfeat, c1, c2, c3, k)
SA_WINDOW_SHAPES = {s for s in _SA_BF16_SHAPES if s[0] <= 3}


def sa_scale_bf16_supported(cfeat, layers, k) -> bool:
    return (mlp_dtype() == "bf16" and len(layers) == 3
            and (cfeat, layers[0].cout, layers[1].cout, layers[2].cout, k) in _SA_BF16_SHAPES)


def sa_bf16_image(layers, cfeat: int, pre: bool) -> torch.Tensor:
    """The weight image of one SA scale for captra_sa_scale_bf16 (fragment-ordered bf16 weights + fp32 biases), built once
    on the device and cached with the scale's first layer."""
    l1, l2, l3 = layers
    key = ("sa_img", cfeat, pre, id(l2), id(l3))
    cache = l1._bf16
    if key not in cache:
        nbytes = L.lib().captra_sa_bf16_image_bytes(cfeat, l1.cout, l2.cout, l3.cout)
        img = torch.empty(nbytes, dtype=torch.uint8, device=l1.wt.device)
        with torch.cuda.device(l1.wt.device):
            L.call("captra_pack_sa_bf16", cfeat, l1.cout, l2.cout, l3.cout, 1 if pre else 0, L.ptr(l1.wt), L.ptr(l1.bias),
                   L.ptr(l2.wt), L.ptr(l2.bias), L.ptr(l3.wt), L.ptr(l3.bias), L.ptr(img))
        cache[key] = (img, l2, l3)          # the key holds ids: keep the layers it was built from alive with it
    return cache[key][0]


def sa_scale_bf16(feat, xyz_cn, new_xyz_n3, idx, layers, out, co_off):
    """One SA scale with bf16 MFMA operands (captra_sa_scale_bf16); wide inputs go through the pre-transformed first layer
    (point-major, once per source point)."""
    l1, l2, l3 = layers
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    cfeat = 0 if feat is None else feat.shape[1]
    pre = cfeat + 3 > 6
    if pre:
        lead = l1.leading_rows(cfeat)
        src = torch.empty(B, N, l1.cout, dtype=torch.float32, device=feat.device)
        L.require_device(feat)
        with torch.cuda.device(feat.device):
            # (the LDS-tiled kernel with an fp32 point-major output was measured here and lost: 17.2 us against 13.5 us per launch
            # at 32 clouds -- three K-chunks of 512 points are three exposed staging latencies; the streaming kernel prefetches deeper)
            L.call("captra_pointwise_mlp_bf16_pm", B, cfeat, l1.cout, N, L.ptr(feat), L.ptr(lead.bf16(0, cfeat)), L.ptr(lead.bias),
                   ACT_NONE, L.ptr(src))
        _work("pointwise_mlp", flops=2.0 * B * cfeat * l1.cout * N, nbytes=4.0 * B * N * (cfeat + l1.cout))
    else:
        src = feat
    img = sa_bf16_image(layers, cfeat, pre)
    L.require_device(src, xyz_cn, new_xyz_n3, idx, out)
    with torch.cuda.device(xyz_cn.device):
        L.call("captra_sa_scale_bf16", B, N, M, K, cfeat, l1.cout, l2.cout, l3.cout, 1 if pre else 0, L.ptr(src), L.ptr(xyz_cn),
               L.ptr(new_xyz_n3), L.ptr(idx), L.ptr(img), L.ptr(out), out.shape[1], co_off)
    _work("sa_scale_fused", flops=2.0 * B * M * K * ((3 if pre else cfeat + 3) * l1.cout + l1.cout * l2.cout + l2.cout * l3.cout),
          nbytes=4.0 * B * ((l1.cout if pre else cfeat) * N + 3 * N + M * K + 3 * M + l3.cout * M))
    return out


def sa_first_layer_pre(feat, lin: PackedLinear):
    """v1 (B,c1,N) = b1 + W1[feature rows] feat: the part of an SA scale's first layer that depends on the source
    point only (the chain's first cfeat steps; the packed buffer's leading rows ARE the feature rows)."""
    assert lin.cin == feat.shape[1] + 3
    return pointwise_mlp(feat, lin.leading_rows(feat.shape[1]), ACT_NONE)


USE_SA_PIPE = True       # SA2 scales on the pipelined kernel (csrc/sa_pipe.hip); False = sa_wave_kernel<..., PRE> (A/B, tests)


# ---- f32x6: fp32-equivalent arithmetic on the bf16 matrix pipe (csrc/sa_x6.hip, csrc/dense_x6.hip) -------------------------------------
# cfg['mlp_dtype'] = "f32x6": every operand of the wide shared-MLP layers is a three-way bf16 split (exact), six bf16 MFMAs per k-step,
# fp32 accumulation.  Not bit-identical to the exact fmaf chain (fp32-roundoff-sized differences: tests/test_x6_gpu.py), 16 / 6 of
# its matrix-pipe rate.  Layers without an f32x6 kernel run the exact fp32 path.
_SA_X6_SHAPES = {(0, 32, 32, 64), (0, 64, 64, 128), (0, 64, 96, 128), (3, 32, 32, 64), (3, 64, 64, 128), (3, 64, 96, 128),
                 (320, 128, 128, 256), (320, 128, 196, 256)}      # csrc/sa_x6.hip SX_CASE list
USE_SA_X6 = os.environ.get("CAPTRA_SA_X6", "1") != "0"


def sa_scale_x6_supported(cfeat, layers, k) -> bool:
    return (USE_SA_X6 and mlp_dtype() == "f32x6" and len(layers) == 3 and k % 32 == 0
            and (cfeat, layers[0].cout, layers[1].cout, layers[2].cout) in _SA_X6_SHAPES)


def sa_x6_image(layers, cfeat: int, pre: bool) -> torch.Tensor:
    """The weight image of one SA scale for captra_sa_scale_x6 (layers 2 / 3 as three-way bf16 split fragments + the fp32 biases),
    built once on the device and cached with the scale's first layer."""
    l1, l2, l3 = layers
    key = ("sa_x6_img", cfeat, pre, id(l2), id(l3))
    cache = l1._bf16
    if key not in cache:
        nbytes = L.lib().captra_sa_x6_image_bytes(cfeat, l1.cout, l2.cout, l3.cout)
        img = torch.empty(nbytes, dtype=torch.uint8, device=l1.wt.device)
        with torch.cuda.device(l1.wt.device):
            L.call("captra_pack_sa_x6", cfeat, l1.cout, l2.cout, l3.cout, None if pre else L.ptr(l1.bias), L.ptr(l2.wt), L.ptr(l2.bias),
                   L.ptr(l3.wt), L.ptr(l3.bias), L.ptr(img))
        cache[key] = (img, l2, l3)          # the key holds ids: keep the layers it was built from alive with it
    return cache[key][0]


def sa_scale_x6(feat, xyz_cn, new_xyz_n3, idx, layers, out, co_off):
    """One SA scale in the f32x6 arithmetic (captra_sa_scale_x6); wide inputs go through the pre-transformed first layer (exact fp32,
    point-major, once per source point)."""
    l1, l2, l3 = layers
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    cfeat = 0 if feat is None else feat.shape[1]
    pre = cfeat + 3 > 6
    src = sa_first_layer_pre_pm(feat, l1) if pre else feat
    img = sa_x6_image(layers, cfeat, pre)
    L.require_device(xyz_cn, new_xyz_n3, idx, out)
    with torch.cuda.device(xyz_cn.device):
        L.call("captra_sa_scale_x6", B, N, M, K, cfeat, l1.cout, l2.cout, l3.cout, 1 if pre else 0, L.ptr(src),
               L.ptr(xyz_cn), L.ptr(new_xyz_n3), L.ptr(idx), L.ptr(l1.wt), L.ptr(img), L.ptr(out), out.shape[1], co_off)
    _work("sa_scale_x6", flops=2.0 * B * M * K * ((3 if pre else cfeat + 3) * l1.cout + l1.cout * l2.cout + l2.cout * l3.cout),
          nbytes=4.0 * B * ((l1.cout if pre else cfeat) * N + 3 * N + M * K + 3 * M + l3.cout * M))
    return out


def sa_scale_pipe_supported(cfeat, layers, m: int, k: int, b: int = 1, n: int = 1) -> bool:
    """The launchers' own -2 conditions, mirrored (captra_sa_scale_pre_pm: tile-aligned positions, 32-bit buffer offsets of the
    point-major v1 and of the position index; captra_pointwise_mlp_pm: 16-byte rows of v1, 32-bit offsets of its input), so that
    a shape outside them takes the captra_sa_scale_pre path instead of raising."""
    if not (USE_SA_PIPE and sa_scale_pre_supported(cfeat, layers, k) and (m * k) % 128 == 0):
        return False
    c1 = layers[0].cout
    return c1 % 4 == 0 and c1 * n * 4 < (1 << 31) and b * m * k < (1 << 31) and cfeat * n * 4 < (1 << 31)


def sa_first_layer_pre_pm(feat, lin: PackedLinear):
    """v1 POINT-major (B,N,c1) = b1 + W1[feature rows] feat (captra_pointwise_mlp_pm): what the pipelined SA kernel gathers
    with 16-byte loads."""
    assert lin.cin == feat.shape[1] + 3
    lead = lin.leading_rows(feat.shape[1])
    L.require_device(feat, lead.wt, lead.bias)
    B, cin, N = feat.shape
    out = torch.empty(B, N, lead.cout, dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        L.call("captra_pointwise_mlp_pm", B, cin, lead.cout, N, L.ptr(feat), L.ptr(lead.wt), L.ptr(lead.bias), ACT_NONE, L.ptr(out))
    _work("pointwise_mlp", flops=2.0 * B * cin * lead.cout * N, nbytes=4.0 * B * N * (cin + lead.cout))
    return out


def sa_scale_pre_pm(v1pm, xyz_cn, new_xyz_n3, idx, layers, out, co_off, cfeat, jobs=None):
    """One SA scale from the point-major pre-transformed first layer (captra_sa_scale_pre_pm).  `jobs`: as sa_scale_fused."""
    l1, l2, l3 = layers
    L.require_device(v1pm, xyz_cn, new_xyz_n3, idx, out)
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    if jobs is not None:
        jobs.append(L.SaScaleJob(1, B, N, M, K, cfeat, l1.cout, l2.cout, l3.cout, L.ptr(v1pm), L.ptr(xyz_cn), L.ptr(new_xyz_n3), L.ptr(idx),
                                 L.ptr(l1.wt), None, L.ptr(l2.wt), L.ptr(l2.bias), L.ptr(l3.wt), L.ptr(l3.bias), L.ptr(out), out.shape[1], co_off))
    else:
        with torch.cuda.device(xyz_cn.device):
            L.call("captra_sa_scale_pre_pm", B, N, M, K, cfeat, l1.cout, l2.cout, l3.cout, L.ptr(v1pm), L.ptr(xyz_cn),
                   L.ptr(new_xyz_n3), L.ptr(idx), L.ptr(l1.wt), L.ptr(l2.wt), L.ptr(l2.bias), L.ptr(l3.wt), L.ptr(l3.bias),
                   L.ptr(out), out.shape[1], co_off)
    _work("sa_scale_fused", flops=2.0 * B * M * K * (3 * l1.cout + l1.cout * l2.cout + l2.cout * l3.cout),
          nbytes=4.0 * B * (l1.cout * N + 3 * N + M * K + 3 * M + l3.cout * M))
    return out


def sa_scale_pre(v1, xyz_cn, new_xyz_n3, idx, layers, out, co_off, cfeat):
    """One SA scale from the pre-transformed first layer (captra_sa_scale_pre)."""
    l1, l2, l3 = layers
    L.require_device(v1, xyz_cn, new_xyz_n3, idx, out)
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    with torch.cuda.device(xyz_cn.device):
        L.call("captra_sa_scale_pre", B, N, M, K, cfeat, l1.cout, l2.cout, l3.cout, L.ptr(v1), L.ptr(xyz_cn),
               L.ptr(new_xyz_n3), L.ptr(idx), L.ptr(l1.wt), L.ptr(l2.wt), L.ptr(l2.bias), L.ptr(l3.wt), L.ptr(l3.bias),
               L.ptr(out), out.shape[1], co_off)
    _work("sa_scale_fused", flops=2.0 * B * M * K * (3 * l1.cout + l1.cout * l2.cout + l2.cout * l3.cout),
          nbytes=4.0 * B * (l1.cout * N + 3 * N + M * K + 3 * M + l3.cout * M))
    return out


USE_ROT_READOUT = True   # tracking: rotation read-out (normalise, masked mean, frame, compose) as one launch


def rot_pool_compose(raw, labels_i32, prev_rot, sym: bool, want_delta: bool = False):
    """raw: rotation-head outputs, (B*P,P,R,N) (every head on every cloud) or (B*P,R,N) (head p on cloud (b,p) only);
    labels (B,N) int32, prev_rot (B,P,3,3) -> rotation (B,P,3,3) = prev_rot @ dR (captra_rot_pool_compose);
    with want_delta also dR."""
    L.require_device(raw, labels_i32, prev_rot)
    B, P = prev_rot.shape[:2]
    diag = raw.dim() == 3
    R, N = raw.shape[-2:]
    assert raw.shape[0] == B * P and (diag or raw.shape[1] == P) and R == (3 if sym else 6)
    assert labels_i32.shape == (B, N) and labels_i32.dtype == torch.int32
    assert prev_rot.shape == (B, P, 3, 3) and raw.is_contiguous() and prev_rot.is_contiguous() and labels_i32.is_contiguous()
    rot = torch.empty(B, P, 3, 3, dtype=torch.float32, device=raw.device)
    delta = torch.empty_like(rot) if want_delta else None
    with torch.cuda.device(raw.device):
        L.call("captra_rot_pool_compose", B, P, N, 1 if sym else 0, 1 if diag else 0, L.ptr(raw), L.ptr(labels_i32),
               L.ptr(prev_rot), L.ptr(rot), L.ptr(delta))
    return (rot, delta) if want_delta else rot


USE_GN_FUSED = True      # Conv -> GroupNorm -> ReLU chains: statistics in the conv's epilogue, normalisation in the next conv's load


def gn_chain_supported(x, cout: int) -> bool:
    """A conv whose output is group-normalised can emit the statistics itself when it runs in the 64x64 wave-tile
    configuration of the direct kernel (csrc/pointwise_mlp.hip captra_pointwise_mlp_gn; fp32 path -- and the f32x6 mode, whose
    layers outside csrc/dense_x6.hip's shapes run these exact kernels)."""
    B, cin = x.shape[0], x.shape[1]
    l = x.numel() // max(B * cin, 1)
    return USE_GN_FUSED and mlp_dtype() in ("fp32", "f32x6") and cout > 64 and cin * l * 4 < (1 << 31)


USE_DENSE_X6 = os.environ.get("CAPTRA_DENSE_X6", "1") != "0"


def dense_x6_supported(cin: int, cout: int, l: int) -> bool:
    """Shapes of captra_pointwise_mlp_x6 (csrc/dense_x6.hip: 256-channel x 128-position workgroup tiles, 16-wide k-steps)."""
    return (USE_DENSE_X6 and mlp_dtype() == "f32x6" and cin % 16 == 0 and cin <= 1024 and cout % 256 == 0 and l % 128 == 0 and l > 0)


def dense_x6_image(lin: PackedLinear) -> torch.Tensor:
    """Three-way bf16 split fragment image of a layer for captra_pointwise_mlp_x6, built once and cached with the layer."""
    key = ("dense_x6",)
    if key not in lin._bf16:
        img = torch.empty(L.lib().captra_dense_x6_image_bytes(lin.cin, lin.cout), dtype=torch.uint8, device=lin.wt.device)
        with torch.cuda.device(lin.wt.device):
            L.call("captra_pack_dense_x6", lin.cin, lin.cout, L.ptr(lin.wt), L.ptr(img))
        lin._bf16[key] = img
    return lin._bf16[key]


def pointwise_mlp_gn(x, lin: PackedLinear, ab_in=None, act: int = ACT_NONE, want_stats: bool = False):
    """Dense layer of a Conv -> GroupNorm -> ReLU chain: x (B,cin,L) raw output of the previous layer with its GroupNorm
    coefficients ab_in (B,cin,2) (or a plain input, ab_in None) -> y (B,cout,L) [, stats (B,cout,T,2)]."""
    L.require_device(x, lin.wt, lin.bias, ab_in)
    B, cin = x.shape[0], x.shape[1]
    assert lin.cin == cin and x.is_contiguous()
    l = x.numel() // max(B * cin, 1)
    y = torch.empty((B, lin.cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    if dense_x6_supported(cin, lin.cout, l):
        # cfg['mlp_dtype'] = "f32x6": six bf16 MFMAs per k-step on three-way splits; statistics per 128 positions
        t = l // 128
        stats = torch.empty(B, lin.cout, t, 2, dtype=torch.float32, device=x.device) if want_stats else None
        with torch.cuda.device(x.device):
            L.call("captra_pointwise_mlp_x6", B, cin, lin.cout, l, L.ptr(x), L.ptr(dense_x6_image(lin)), L.ptr(lin.bias), L.ptr(ab_in), act,
                   L.ptr(y), L.ptr(stats), t)
        _work("pointwise_mlp_x6", flops=2.0 * B * cin * lin.cout * l, nbytes=4.0 * B * l * (cin + lin.cout))
        return (y, stats) if want_stats else y
    o = L.current_opts()
    t = L.lib().captra_pointwise_mlp_gn_tiles_ex(B, lin.cout, l, C.byref(o) if o is not None else None)      # 64- or 32-position statistics tiles, by launch shape
    stats = torch.empty(B, lin.cout, t, 2, dtype=torch.float32, device=x.device) if want_stats else None
    with torch.cuda.device(x.device):
        L.call("captra_pointwise_mlp_gn", B, cin, lin.cout, l, L.ptr(x), L.ptr(lin.wt), L.ptr(lin.bias), L.ptr(ab_in), act,
               L.ptr(y), L.ptr(stats), t)
    _work("pointwise_mlp", flops=2.0 * B * cin * lin.cout * l, nbytes=4.0 * B * l * (cin + lin.cout))
    return (y, stats) if want_stats else y


def gn_finalize(stats, num_groups: int, gamma, beta, eps: float, n: int, tile_major: bool = False):
    """stats (B,C,T,2) -- or (B,T,C,2) when tile_major -- partial (sum, sum of squares) -> ab (B,C,2) with GroupNorm(x) = a*x + b."""
    L.require_device(stats, gamma, beta)
    if tile_major:
        B, T, C, _ = stats.shape
    else:
        B, C, T, _ = stats.shape
    ab = torch.empty(B, C, 2, dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        L.call("captra_gn_finalize_tm" if tile_major else "captra_gn_finalize", B, C, C // num_groups, T, n, float(eps),
               L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(ab))
    return ab


USE_MLP_CHAIN = True     # three dense layers in one launch where the shape is instantiated (else layer by layer)
_CHAIN3_SHAPES = {(134, 128, 128, 128), (131, 128, 128, 128)}   # csrc/mlp_chain.hip CHAIN_CASE list


def mlp_chain3(x, layers, act3: int = ACT_RELU):
    """x (B,c0,*) -> act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3) (B,c3,*): one launch when (c0,c1,c2,c3) is an
    instantiated shape of captra_mlp_chain3, otherwise three captra_pointwise_mlp launches (same bits)."""
    assert len(layers) == 3
    shape = (layers[0].cin, layers[0].cout, layers[1].cout, layers[2].cout)
    assert x.shape[1] == shape[0] and layers[1].cin == shape[1] and layers[2].cin == shape[2], (x.shape, shape)
    B = x.shape[0]
    l = x.numel() // max(B * shape[0], 1)
    if mlp_dtype() == "bf16":
        return mlp_chain_bf16(x, layers, [ACT_RELU, ACT_RELU, act3])
    # under split_k with few positions: three split-k launches (each fills the chip) instead of the one-launch chain, whose waves own
    # 32 positions each -- 128 waves for one 4096-point cloud, every one a serial chain of the three layers (49 -> ~25 us)
    few = B * l <= _split_k_limit()
    if not few and chain_x6_supported(shape[0], [lin.cout for lin in layers], B * l):
        return mlp_chain3_x6(x, layers, act3)
    if few or not (USE_MLP_CHAIN and exact_path() and shape in _CHAIN3_SHAPES and shape[0] * l * 4 < (1 << 31)):
        y = pointwise_mlp(x, layers[0], ACT_RELU)
        y = pointwise_mlp(y, layers[1], ACT_RELU)
        return pointwise_mlp(y, layers[2], act3)
    L.require_device(x, *(t for lin in layers for t in (lin.wt, lin.bias)))
    out = torch.empty((B, shape[3]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_mlp_chain3", B, shape[0], shape[1], shape[2], shape[3], l, L.ptr(x),
               L.ptr(layers[0].wt), L.ptr(layers[0].bias), L.ptr(layers[1].wt), L.ptr(layers[1].bias),
               L.ptr(layers[2].wt), L.ptr(layers[2].bias), act3, L.ptr(out))
    _work("mlp_chain3", flops=2.0 * B * l * (shape[0] * shape[1] + shape[1] * shape[2] + shape[2] * shape[3]),
          nbytes=4.0 * B * l * (shape[0] + shape[3]))
    return out


USE_COORD_TAIL = True
_COORD_TAIL_SHAPES = {(134, 2, 3), (134, 4, 12), (134, 3, 9), (134, 2, 6)}   # csrc/mlp_chain.hip TAIL_CASE list


USE_CHAIN_X6 = os.environ.get("CAPTRA_CHAIN_X6", "1") != "0"
CHAIN_X6_MIN_POSITIONS = 16384        # below: the exact kernels (a chain wave owns 32 positions: few positions leave the chip idle either way)


def chain_x6_supported(c0: int, widths, positions: int) -> bool:
    """The f32x6 chain kernels (csrc/chain_x6.hip): c0 in {131, 134} -> 128 -> 128 -> 128 [+ the CoordinateNet heads]."""
    return (USE_CHAIN_X6 and mlp_dtype() == "f32x6" and c0 in (131, 134) and positions >= CHAIN_X6_MIN_POSITIONS
            and all(w == 128 for w in widths[:3]))


def chain_x6_image(layers) -> torch.Tensor:
    """The weight image of a chain for captra_mlp_chain3_x6 / captra_coord_tail_x6 (every layer's split fragment triples, then 128
    fp32 bias slots per layer), built once on the device and cached with the chain's first layer."""
    key = ("chain_x6_img",) + tuple(id(lin) for lin in layers[1:])
    cache = layers[0]._bf16
    if key not in cache:
        nl = len(layers)
        cin = (C.c_int * nl)(*[lin.cin for lin in layers])
        cout = (C.c_int * nl)(*[lin.cout for lin in layers])
        L.lib().captra_chain_x6_image_bytes.restype = C.c_longlong
        nbytes = L.lib().captra_chain_x6_image_bytes(nl, cin, cout)
        frag_total = nbytes - nl * 128 * 4
        img = torch.empty(nbytes, dtype=torch.uint8, device=layers[0].wt.device)
        off = 0
        with torch.cuda.device(img.device):
            for i, lin in enumerate(layers):
                L.call("captra_pack_chain_x6", i, lin.cin, lin.cout, 1 if i == 0 else 0, off, frag_total, L.ptr(lin.wt), L.ptr(lin.bias), L.ptr(img))
                off += ((lin.cout + 31) // 32) * ((((lin.cin + 15) // 16) * 3 + 3) // 4 * 4) * 1024
        assert off == frag_total
        cache[key] = (img, list(layers))
    return cache[key][0]


def mlp_chain3_x6(x, layers, act3: int = ACT_RELU):
    """mlp_chain3 in the f32x6 arithmetic (captra_mlp_chain3_x6)."""
    B, c0 = x.shape[:2]
    l = x.numel() // max(B * c0, 1)
    img = chain_x6_image(layers)
    L.require_device(x)
    out = torch.empty((B, 128) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.call("captra_mlp_chain3_x6", B, c0, l, L.ptr(x), L.ptr(img), act3, L.ptr(out))
    _work("mlp_chain3_x6", flops=2.0 * B * l * (c0 * 128 + 2 * 128 * 128), nbytes=4.0 * B * l * (c0 + 128))
    return out


def coord_tail_supported(x, layers) -> bool:
    """layers = [fp1a, fp1b, conv1, seg, nocs hidden, nocs out] (PackedLinear)."""
    if not (USE_COORD_TAIL and (exact_path() or mlp_dtype() == "f32x6") and len(layers) == 6):
        return False
    l = x.numel() // max(x.shape[0] * x.shape[1], 1)
    widths_ok = all(lin.cout == 128 for lin in (layers[0], layers[1], layers[2], layers[4])) and all(lin.cin == 128 for lin in layers[1:])
    return widths_ok and (x.shape[1], layers[3].cout, layers[5].cout) in _COORD_TAIL_SHAPES and x.shape[1] * l * 4 < (1 << 31)


def coord_tail(x, layers, nocs_act: int = ACT_SIGMOID_M05):
    """x (B,c0,N) -> (seg logits (B,S,N), nocs (B,3P,N)): FP1 MLP + conv1 + both CoordNet heads in one launch."""
    import ctypes
    L.require_device(x, *(t for lin in layers for t in (lin.wt, lin.bias)))
    B, c0 = x.shape[:2]
    l = x.numel() // max(B * c0, 1)
    seg = torch.empty((B, layers[3].cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    nocs = torch.empty((B, layers[5].cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    if chain_x6_supported(c0, [lin.cout for lin in layers], B * l) and not B * l <= _split_k_limit():
        img = chain_x6_image(layers)
        with torch.cuda.device(x.device):
            L.call("captra_coord_tail_x6", B, c0, layers[3].cout, layers[5].cout, l, L.ptr(x), L.ptr(img), nocs_act, L.ptr(seg), L.ptr(nocs))
        _work("coord_tail_x6", flops=2.0 * B * l * sum(lin.cin * lin.cout for lin in layers), nbytes=4.0 * B * l * (c0 + layers[3].cout + layers[5].cout))
        return seg, nocs
    wp = (ctypes.c_void_p * 6)(*[lin.wt.data_ptr() for lin in layers])
    bp = (ctypes.c_void_p * 6)(*[lin.bias.data_ptr() for lin in layers])
    with torch.cuda.device(x.device):
        L.call("captra_coord_tail", B, c0, layers[3].cout, layers[5].cout, l, L.ptr(x), wp, bp, nocs_act, L.ptr(seg), L.ptr(nocs))
    _work("coord_tail", flops=2.0 * B * l * sum(lin.cin * lin.cout for lin in layers), nbytes=4.0 * B * l * (c0 + layers[3].cout + layers[5].cout))
    return seg, nocs


def sa_group_mlp(feat, xyz_cn, new_xyz_n3, idx, lin: PackedLinear):
    """First SA layer with group + centre-subtract + concat fused into the load -> (B,cout,M,K)."""
    wt, bias = lin.wt, lin.bias
    L.require_device(feat, xyz_cn, new_xyz_n3, idx, wt, bias)
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    cfeat = 0 if feat is None else feat.shape[1]
    cout = lin.cout
    assert lin.cin == cfeat + 3
    y = torch.empty(B, cout, M, K, dtype=torch.float32, device=xyz_cn.device)
    with torch.cuda.device(xyz_cn.device):
        L.call("captra_sa_group_mlp", B, N, M, K, cfeat, cout, L.ptr(feat), L.ptr(xyz_cn), L.ptr(new_xyz_n3), L.ptr(idx),
               L.ptr(wt), L.ptr(bias), L.ptr(y))
    _work("sa_group_mlp", flops=2.0 * B * (cfeat + 3) * cout * M * K,
          nbytes=4.0 * B * ((cfeat + 3) * N + M * K + 3 * M + cout * M * K))
    return y


def mlp_max(x, lin: PackedLinear, out, co_off: int):
    """Last SA layer + max over K: x (B,cin,M,K) -> out[:, co_off:co_off+cout, :] (out is (B,Ctot,M))."""
    wt, bias = lin.wt, lin.bias
    L.require_device(x, wt, bias, out)
    B, cin, M, K = x.shape
    cout = lin.cout
    assert lin.cin == cin
    if mlp_dtype() == "bf16":     # bf16 dense layer, then the (exact) max: only SA3's 128-point layer takes this route
        out[:, co_off:co_off + cout, :] = pointwise_mlp(x, lin, ACT_RELU).max(dim=3)[0]
        return out
    with torch.cuda.device(x.device):
        L.call("captra_mlp_max", B, cin, cout, M, K, L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(out), out.shape[1], co_off)
    _work("mlp_max", flops=2.0 * B * cin * cout * M * K, nbytes=4.0 * B * (cin * M * K + cout * M))
    return out


USE_SA_FUSED = True   # one launch per SA scale (sa_fused.hip); False = layer-by-layer kernels (A/B, tests)


def sa_scale_fusable(k: int, layers) -> bool:
    return USE_SA_FUSED and len(layers) == 3 and k % 32 == 0 and 128 % k == 0 and all(l.cout <= 256 for l in layers)


def sa_scales_multi(jobs, device) -> None:
    """A level's recorded scales (the `jobs=` lists of sa_scale_fused / sa_scale_pre_pm) in ONE call: captra_sa_scales_multi runs
    them as one launch where they are the level's shapes in order, else one after the other -- the job table is this call's
    argument, nothing is remembered between calls."""
    arr = (L.SaScaleJob * len(jobs))(*jobs)
    o = L.current_opts()
    with torch.cuda.device(device):
        L.check(L.lib().captra_sa_scales_multi(len(jobs), C.cast(arr, C.c_void_p), C.byref(o) if o is not None else None, L.stream_ptr()),
                "captra_sa_scales_multi")


def sa_scale_fused(feat, xyz_cn, new_xyz_n3, idx, layers, out, co_off: int, jobs=None):
    """Whole SA scale in one launch: layers = [PackedLinear] * 3; writes out[:, co_off:co_off+c3, :].  `jobs` (a list): the call is
    appended as a job for sa_scales_multi instead of launched."""
    L.require_device(feat, xyz_cn, new_xyz_n3, idx, out, *[t for l in layers for t in (l.wt, l.bias)])
    B, _, N = xyz_cn.shape
    _, M, K = idx.shape
    cfeat = 0 if feat is None else feat.shape[1]
    l1, l2, l3 = layers
    (w1, b1), (w2, b2), (w3, b3) = (l1.wt, l1.bias), (l2.wt, l2.bias), (l3.wt, l3.bias)
    c1, c2, c3 = l1.cout, l2.cout, l3.cout
    assert l1.cin == cfeat + 3 and l2.cin == c1 and l3.cin == c2
    if jobs is not None:
        jobs.append(L.SaScaleJob(0, B, N, M, K, cfeat, c1, c2, c3, L.ptr(feat), L.ptr(xyz_cn), L.ptr(new_xyz_n3), L.ptr(idx), L.ptr(w1), L.ptr(b1),
                                 L.ptr(w2), L.ptr(b2), L.ptr(w3), L.ptr(b3), L.ptr(out), out.shape[1], co_off))
    else:
        with torch.cuda.device(xyz_cn.device):
            L.call("captra_sa_scale_fused", B, N, M, K, cfeat, c1, c2, c3, L.ptr(feat), L.ptr(xyz_cn), L.ptr(new_xyz_n3),
                   L.ptr(idx), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(w3), L.ptr(b3), L.ptr(out), out.shape[1], co_off)
    _work("sa_scale_fused", flops=2.0 * B * M * K * ((cfeat + 3) * c1 + c1 * c2 + c2 * c3),
          nbytes=4.0 * B * ((cfeat + 3) * N + M * K + 3 * M + c3 * M))
    return out


def three_nn_weights(unknown_n3, known_n3):
    """3-NN + normalised inverse-distance weights -> (idx (B,N,3) int32, weight (B,N,3))."""
    L.require_device(unknown_n3, known_n3)
    B, N, _ = unknown_n3.shape
    S = known_n3.shape[1]
    idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown_n3.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=unknown_n3.device)
    with torch.cuda.device(unknown_n3.device):
        L.call("captra_three_nn_weights", B, N, S, L.ptr(unknown_n3), L.ptr(known_n3), L.ptr(idx), L.ptr(w))
    _work("three_nn_weights", nbytes=B * (12.0 * N + 12.0 * S + 24.0 * N))
    return idx, w


def interp_concat(skip, feat_known, idx, weight):
    """cat([skip (B,c1,N), sum_j w_j feat_known (B,c2,S)[:, idx_j]]) -> (B,c1+c2,N)."""
    L.require_device(skip, feat_known, idx, weight)
    B, N, _ = idx.shape
    c2, S = feat_known.shape[1], feat_known.shape[2]
    c1 = 0 if skip is None else skip.shape[1]
    out = torch.empty(B, c1 + c2, N, dtype=torch.float32, device=feat_known.device)
    with torch.cuda.device(feat_known.device):
        L.call("captra_interp_concat", B, N, S, c1, c2, L.ptr(skip), L.ptr(feat_known), L.ptr(idx), L.ptr(weight), L.ptr(out))
    _work("interp_concat", nbytes=4.0 * B * (c1 * N + c2 * S + 6 * N + (c1 + c2) * N))
    return out


def fp_interpolate_concat(unknown_n3, known_n3, skip, feat_known, nn=None):
    """three_nn + inverse-distance weights + interpolate + cat([skip, interp]) -> (B,c1+c2,N).
    `nn` = (idx, weight) from three_nn_weights, when another network already computed them."""
    if nn is None:
        nn = three_nn_weights(unknown_n3, known_n3)
    return interp_concat(skip, feat_known, nn[0], nn[1])


def group_norm_relu(x, num_groups: int, gamma, beta, eps: float, relu: bool = True):
    """GroupNorm(num_groups) + optional ReLU over x (B,C,N) in one pass."""
    L.require_device(x, gamma, beta)
    B, C, N = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        L.call("captra_group_norm_relu", B, C, N, C // num_groups, float(eps), 1 if relu else 0, L.ptr(x), L.ptr(gamma),
               L.ptr(beta), L.ptr(y))
    _work("group_norm_relu", nbytes=8.0 * B * C * N)
    return y
