"""Parity of the ten pointnet2_cuda operators (HIP, through the C ABI) against the CPU oracle.

Bit-exact for index outputs (FPS, ball query, three_nn idx, knn idx) and for pure copies
(group, gather); exact-equal for squared distances and interpolation as well, since both sides
evaluate the same separately-rounded fp32 expression.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from tests import clouds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pn(device):
    from captra_amd.pointnet_lib import pointnet2_utils as pn
    return pn


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _nocs_batch(ids, dup=False):
    fn = clouds.s_nocs_dup if dup else clouds.s_nocs
    return np.stack([fn(i)[0] for i in ids]).astype(np.float32)


# ------------------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize("n,m", [(4096, 512), (512, 128), (4096, 1), (100, 37), (64, 64), (1000, 300), (5000, 64)])
def test_fps_uniform(pn, device, n, m):
    rng = np.random.default_rng(n * 7 + m)
    xyz = (rng.random((3, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    got = pn.furthest_point_sample(_dev(xyz, device), m).cpu().numpy()
    ref = O.furthest_point_sample(xyz, m)
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, ref)


def test_fps_nocs_and_duplicates(pn, device):
    xyz = np.concatenate([_nocs_batch(range(4)), _nocs_batch(range(4), dup=True)])
    got = pn.furthest_point_sample(_dev(xyz, device), 512).cpu().numpy()
    np.testing.assert_array_equal(got, O.furthest_point_sample(xyz, 512))


def test_fps_all_points_identical(pn, device):
    xyz = np.full((2, 777, 3), 0.25, np.float32)
    got = pn.furthest_point_sample(_dev(xyz, device), 16).cpu().numpy()
    np.testing.assert_array_equal(got, O.furthest_point_sample(xyz, 16))
    assert (got == 0).all()  # every distance is 0: the lowest index wins every round


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("waves", [1, 2, 4, 8, 16])
def test_fps_every_wave_configuration(pn, device, waves, variant):
    """Every workgroup shape of both register-resident kernels (variant 0: blocked ownership + ballot pick + packed
    math, the default; variant 1: strided ownership + two DPP reductions) on clean and duplicated clouds."""
    import ctypes
    from captra_amd import _lib
    xyz = np.concatenate([_nocs_batch(range(2)), _nocs_batch(range(1), dup=True)])
    ref = O.furthest_point_sample(xyz, 128)
    _lib.lib().captra_fps_set_waves(ctypes.c_int(waves))
    _lib.lib().captra_fps_set_variant(ctypes.c_int(variant))
    try:
        got = pn.furthest_point_sample(_dev(xyz, device), 128).cpu().numpy()
    finally:
        _lib.lib().captra_fps_set_waves(ctypes.c_int(0))
        _lib.lib().captra_fps_set_variant(ctypes.c_int(0))
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n,m", [(4096, 512), (512, 128), (1000, 300), (130, 130), (3, 2)])
def test_fps_first_generation_kernel(pn, device, n, m):
    """The strided-ownership kernel stays selectable (captra_fps_set_variant): same indices."""
    import ctypes
    from captra_amd import _lib
    rng = np.random.default_rng(n + m)
    xyz = (rng.random((2, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    _lib.lib().captra_fps_set_variant(ctypes.c_int(1))
    try:
        got = pn.furthest_point_sample(_dev(xyz, device), m).cpu().numpy()
    finally:
        _lib.lib().captra_fps_set_variant(ctypes.c_int(0))
    np.testing.assert_array_equal(got, O.furthest_point_sample(xyz, m))


@pytest.mark.parametrize("n,m", [(4096, 512), (512, 128), (1000, 300), (77, 77), (3, 2), (12000, 40), (20480, 64), (32768, 8)])
def test_fps_gather_one_launch(device, n, m):
    """captra_fps_gather: indices of furthest_point_sample + the sampled coordinates in both layouts."""
    from captra_amd import fused
    rng = np.random.default_rng(n * 3 + m)
    xyz = (rng.random((2, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    idx, n3, cn = fused.fps_gather(_dev(xyz, device), m)
    ref = O.furthest_point_sample(xyz, m)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    picked = np.take_along_axis(xyz, ref[..., None].astype(np.int64).repeat(3, -1), axis=1)
    np.testing.assert_array_equal(n3.cpu().numpy(), picked)
    np.testing.assert_array_equal(cn.cpu().numpy(), picked.transpose(0, 2, 1))
    assert fused.fps_gather(_dev(np.zeros((1, 40000, 3), np.float32), device), 4) is None   # too large: caller falls back


def _surface_cloud(seed, n):
    """Points on a noisy cylinder + clutter (the shape of a depth crop), float32."""
    rng = np.random.default_rng(seed)
    th, h = rng.random(n) * 2 * np.pi, rng.random(n) - 0.5
    pts = np.stack([0.2 * np.cos(th), h, 0.2 * np.sin(th)], -1) + rng.normal(0, 0.004, (n, 3))
    pts[: n // 5] = rng.random((n // 5, 3)) - 0.5
    return rng.permutation(pts).astype(np.float32)


@pytest.mark.parametrize("n,m", [(8192, 700), (12288, 300), (16384, 2048), (20480, 4096), (9001, 513)])
def test_fps_pruned_kernel_same_picks(pn, device, n, m):
    """Clouds of 8k-20k points take the spatially pruned kernel (csrc/fps_pruned.hip): picks, sampled coordinates and
    the running-minimum array it hands back are those of the plain kernel and of the oracle, on a surface-like cloud
    and on a uniform one."""
    import ctypes
    from captra_amd import _lib, fused
    xyz = np.stack([_surface_cloud(n + m, n), clouds.s_uni(n, n)])
    ref = O.furthest_point_sample(xyz, m)
    x = _dev(xyz, device)
    got = pn.furthest_point_sample(x, m).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    idx, n3, cn = fused.fps_gather(x, m)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    picked = np.take_along_axis(xyz, ref[..., None].astype(np.int64).repeat(3, -1), axis=1)
    np.testing.assert_array_equal(n3.cpu().numpy(), picked)
    np.testing.assert_array_equal(cn.cpu().numpy(), picked.transpose(0, 2, 1))
    # the drop-in op's temp array (in: 1e10, out: final running minima) against the unpruned kernel
    temp_p = torch.full((2, n), 1e10, device=device)
    out_p = torch.zeros(2, m, dtype=torch.int32, device=device)
    _lib.call("captra_furthest_point_sampling", 2, n, m, x.data_ptr(), temp_p.data_ptr(), out_p.data_ptr())
    _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(0))
    try:
        temp_u = torch.full((2, n), 1e10, device=device)
        out_u = torch.zeros(2, m, dtype=torch.int32, device=device)
        _lib.call("captra_furthest_point_sampling", 2, n, m, x.data_ptr(), temp_u.data_ptr(), out_u.data_ptr())
    finally:
        _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(8192))
    assert torch.equal(out_p, out_u) and torch.equal(temp_p, temp_u)


def test_fps_pruned_duplicate_padded_cloud(device):
    """The on-the-fly crop doubles a short candidate list until it holds num_points (nocs_data_process.py:105-106):
    every point then exists twice or four times, every maximum is attained several times, and once all distinct points
    are taken the sampler keeps returning index 0.  The pruned kernel resolves those ties by ORIGINAL index."""
    from captra_amd import fused
    base = _surface_cloud(5, 2500)
    xyz = np.concatenate([base] * 4)[None]                     # 10000 points, 2500 distinct
    ref = O.furthest_point_sample(xyz, 4096)
    idx, _, _ = fused.fps_gather(_dev(xyz, device), 4096)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    assert (ref[0, 2500:] == 0).all() and len(set(ref[0, :2500])) == 2500
    # degenerate: every point the same (zero-extent bounding box, one Morton cell, every distance 0) -> index 0 forever
    same = np.tile(np.array([[0.3, -0.2, 0.9]], np.float32), (1, 8193, 1))
    idx, n3, _ = fused.fps_gather(_dev(same, device), 40)
    assert (idx.cpu().numpy() == 0).all() and np.array_equal(n3.cpu().numpy(), same[:, :40])
    # two distinct points among thousands of copies of one of them, and a point count that is not a multiple of 64
    mixed = same.copy()
    mixed[0, 5000] = (0.4, -0.2, 0.9)
    np.testing.assert_array_equal(fused.fps_gather(_dev(mixed, device), 5)[0].cpu().numpy(), O.furthest_point_sample(mixed, 5))


@pytest.mark.parametrize("kind", ["lattice", "coarse_grid", "two_clusters", "line"])
def test_fps_pruned_many_picks_per_round_under_ties(device, kind):
    """The pruned sampler certifies up to four picks per round (csrc/fps_pruned.hip header): a candidate must be positive,
    untouched by the picks before it and strictly above what their waves still hold.  Clouds built to stress exactly
    that: a lattice (all points distinct, almost every distance shared by many pairs), a coarse grid (1728 distinct
    positions, each held ~6 times: every maximum attained several times, then all-zero), two far-apart clusters (the
    first picks alternate between them: candidates far from every earlier pick), points on a line (one Morton axis)."""
    from captra_amd import _lib, fused
    rng = np.random.default_rng(11)
    if kind == "lattice":
        g = np.arange(21, dtype=np.float32) * 0.05
        pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)              # 9261 distinct points
        xyz, m = rng.permutation(pts)[None], 3000
    elif kind == "coarse_grid":
        xyz, m = (rng.integers(0, 12, (1, 10000, 3)).astype(np.float32) * 0.125), 2500     # more picks than distinct points
    elif kind == "two_clusters":
        a = rng.normal(0, 0.01, (6000, 3)) + (1.0, 0, 0)
        b = rng.normal(0, 0.01, (6000, 3)) - (1.0, 0, 0)
        xyz, m = rng.permutation(np.concatenate([a, b])).astype(np.float32)[None], 1024
    else:
        t = rng.random(9000).astype(np.float32)
        xyz, m = np.stack([t, np.zeros_like(t), np.zeros_like(t)], -1)[None], 2048
    xyz = np.ascontiguousarray(xyz, np.float32)
    ref = O.furthest_point_sample(xyz, m)
    idx, n3, _ = fused.fps_gather(_dev(xyz, device), m)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(n3.cpu().numpy()[0], xyz[0, ref[0]])
    # the drop-in op's running-minimum array (every sample but the last applied, whatever round it was picked in)
    temp = torch.full((1, xyz.shape[1]), 1e10, device=device)
    out = torch.zeros(1, m, dtype=torch.int32, device=device)
    _lib.call("captra_furthest_point_sampling", 1, xyz.shape[1], m, _dev(xyz, device).data_ptr(), temp.data_ptr(), out.data_ptr())
    t_ref = np.full((1, xyz.shape[1]), 1e10, np.float32)
    O.furthest_point_sample(xyz, m, temp=t_ref)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(temp.cpu().numpy(), t_ref)


def test_fps_gather_ragged_batch(device):
    """captra_fps_gather_ragged: clouds padded to a common stride, each sampling from its own prefix — equal to sampling
    every cloud on its own (pruned kernel at 20480 stride, register kernel at 3000)."""
    from captra_amd import fused
    for stride, counts, m in [(20480, [20480, 8300, 4096, 13001], 4096), (3000, [3000, 1500, 64, 2999], 64)]:
        full = np.stack([_surface_cloud(40 + i, stride) for i in range(len(counts))])
        ns = torch.tensor(counts, dtype=torch.int32, device=device)
        idx, n3, cn = fused.fps_gather(_dev(full, device), m, n_per_cloud=ns)
        for i, c in enumerate(counts):
            ref = O.furthest_point_sample(full[i:i + 1, :c], m)[0]
            np.testing.assert_array_equal(idx[i].cpu().numpy(), ref)
            np.testing.assert_array_equal(n3[i].cpu().numpy(), full[i, ref])


def test_fps_dispatcher_is_never_far_behind_the_better_kernel(device):
    """the plain / pruned dispatch (csrc/fps.hip: clouds of >= 8192 points take the pruned kernel) over the
    grid of tools/bench_fps.py -- surface-like and uniform clouds around the crossover -- must stay within 10 % of whichever
    kernel is faster (best of three timings each; same picks from both)."""
    import ctypes
    import time
    from captra_amd import _lib, fused
    lib = _lib.lib()

    def best(x, m, pm):
        lib.captra_fps_set_pruned_min(ctypes.c_int(pm))
        out = fused.fps_gather(x, m)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = fused.fps_gather(x, m)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts), out[0]

    try:
        for n, m, B in [(8192, 1024, 8), (12288, 2048, 8), (16384, 2048, 8), (15000, 4096, 16)]:
            for kind in ("surface", "uniform"):
                rng = np.random.default_rng(n + len(kind))
                if kind == "uniform":
                    xyz = np.stack([clouds.s_uni(i, n) for i in range(B)])
                else:
                    th, hh = rng.random((B, n)) * 2 * np.pi, rng.random((B, n)) - 0.5
                    xyz = (np.stack([0.2 * np.cos(th), hh, 0.2 * np.sin(th)], -1) + rng.normal(0, 0.004, (B, n, 3))).astype(np.float32)
                x = torch.from_numpy(np.ascontiguousarray(xyz, np.float32)).to(device)
                t_plain, i_plain = best(x, m, 0)
                t_pruned, i_pruned = best(x, m, 1)
                t_disp, i_disp = best(x, m, 8192)                     # the library's default threshold
                assert torch.equal(i_plain, i_pruned) and torch.equal(i_disp, i_plain)
                assert t_disp <= 1.10 * min(t_plain, t_pruned), (n, m, kind, t_plain, t_pruned, t_disp)
    finally:
        lib.captra_fps_set_pruned_min(ctypes.c_int(8192))


def test_fps_big_cloud_fallback(pn, device):
    xyz = clouds.s_uni(0, 40000)[None]
    got = pn.furthest_point_sample(_dev(xyz, device), 64).cpu().numpy()
    np.testing.assert_array_equal(got, O.furthest_point_sample(xyz, 64))


def test_fps_temp_is_left_with_min_distances(device):
    from captra_amd import pointnet2_cuda
    xyz = _nocs_batch(range(2))
    B, N, _ = xyz.shape
    temp = torch.full((B, N), 1e10, device=device)
    idx = torch.empty(B, 64, dtype=torch.int32, device=device)
    pointnet2_cuda.furthest_point_sampling_wrapper(B, N, 64, _dev(xyz, device), temp, idx)
    t_ref = np.full((B, N), 1e10, np.float32)
    O.furthest_point_sample(xyz, 64, temp=t_ref)
    np.testing.assert_array_equal(temp.cpu().numpy(), t_ref)


# ----------------------------------------------------------------------------------- ball query
SA_PAIRS = [(0.05, 32), (0.1, 64), (0.2, 128)]


@pytest.mark.parametrize("radius,k", SA_PAIRS)
def test_ball_query_sa1(pn, device, radius, k):
    xyz = _nocs_batch(range(3))
    centres = O.furthest_point_sample(xyz, 512)
    new_xyz = np.take_along_axis(xyz, centres[..., None].astype(np.int64), 1)
    got = pn.ball_query(radius, k, _dev(xyz, device), _dev(new_xyz, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.ball_query(radius, k, xyz, new_xyz))


@pytest.mark.parametrize("radius,k", [(0.2, 64), (0.4, 128)])
def test_ball_query_sa2(pn, device, radius, k):
    xyz = _nocs_batch(range(3))
    l1 = O.furthest_point_sample(xyz, 512)
    xyz1 = np.take_along_axis(xyz, l1[..., None].astype(np.int64), 1)
    l2 = O.furthest_point_sample(xyz1, 128)
    xyz2 = np.take_along_axis(xyz1, l2[..., None].astype(np.int64), 1)
    got = pn.ball_query(radius, k, _dev(xyz1, device), _dev(xyz2, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.ball_query(radius, k, xyz1, xyz2))


@pytest.mark.parametrize("n,m,k,radius", [(1, 1, 4, 0.5), (63, 5, 7, 0.3), (65, 33, 1, 0.2), (1000, 37, 200, 0.6),
                                            (9000, 40, 16, 0.05), (20000, 20, 300, 0.2)])
def test_ball_query_ragged_and_multi_tile(pn, device, n, m, k, radius):
    rng = np.random.default_rng(n + m)
    xyz = (rng.random((2, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    new_xyz = (rng.random((2, m, 3), dtype=np.float32) - 0.5).astype(np.float32)
    got = pn.ball_query(radius, k, _dev(xyz, device), _dev(new_xyz, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.ball_query(radius, k, xyz, new_xyz))


@pytest.mark.parametrize("case", ["uniform", "surface", "outside", "duplicates", "flat", "clustered", "odd_n"])
def test_ball_query_grid_path_vs_oracle(device, case):
    """Opt-in grid path (ball_query.hip PRUNE): clouds of 1024..4096 points answer their small radii from a cell grid: bit-exact lists against
    the oracle, and against the index-order scan of the same library, for centres inside / outside the cloud's box and NaN,
    duplicate-padded clouds (thousands of points in one cell), degenerate extents (a plane; a single location), radii on
    both sides of the grid / scan threshold, up to three radii per scan."""
    import ctypes
    from captra_amd import _lib
    rng = np.random.default_rng(len(case) * 7 + 1)
    B, n, m = 2, 4096, 100
    xyz = (rng.random((B, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    if case == "surface":
        v = rng.standard_normal((B, n, 3)).astype(np.float32)
        xyz = (0.5 * v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    elif case == "duplicates":
        xyz[:, 1500:] = xyz[:, :1]                      # duplicate-padded cloud (reference data_utils pads by repetition)
    elif case == "flat":
        xyz[..., 1] = 0.25
        xyz[1] = xyz[1, :1]                             # second cloud: a single location (extent 0: no grid)
    elif case == "clustered":
        xyz[:, : n // 2] = (0.02 * rng.standard_normal((B, n // 2, 3)) + 0.3).astype(np.float32)
    elif case == "odd_n":
        n = 3001
        xyz = xyz[:, :n].copy()
    pick = rng.integers(0, n, (B, m))
    new_xyz = np.take_along_axis(xyz, pick[..., None], 1).copy()
    if case == "outside":
        new_xyz[:, :30] += rng.choice([-0.7, 0.7, 0.04], (B, 30, 3)).astype(np.float32)
        new_xyz[:, 30] = 50.0
        new_xyz[:, 31] = np.nan
        new_xyz[:, 32, 0] = -0.5 - 0.03                 # just outside the box, ball reaching in
    d_xyz, d_new = _dev(xyz, device), _dev(new_xyz, device)
    lib = _lib.lib()
    for radii, ks in [((0.05, 0.1, 0.2), (32, 64, 128)), ((0.03,), (8,)), ((0.12, 0.6), (300, 16)), ((0.149, 0.151), (64, 64))]:
        nr = len(radii)
        res = {}
        for prune in (1, 0):
            lib.captra_ball_query_set_prune(ctypes.c_int(prune))
            try:
                outs = [torch.full((B, m, k), -1, dtype=torch.int32, device=device) for k in ks]
                _lib.call("captra_ball_query_multi", B, n, m, nr, ctypes.cast((ctypes.c_float * nr)(*radii), ctypes.c_void_p),
                          ctypes.cast((ctypes.c_int * nr)(*ks), ctypes.c_void_p), d_new.data_ptr(), d_xyz.data_ptr(),
                          ctypes.cast((ctypes.c_void_p * nr)(*[o.data_ptr() for o in outs]), ctypes.c_void_p))
                res[prune] = [o.cpu().numpy() for o in outs]
            finally:
                lib.captra_ball_query_set_prune(ctypes.c_int(0))
        for r, k, a, b_ in zip(radii, ks, res[1], res[0]):
            np.testing.assert_array_equal(a, b_, err_msg=f"grid vs scan, r={r}")
            np.testing.assert_array_equal(a, O.ball_query(r, k, xyz, new_xyz), err_msg=f"vs oracle, r={r}")


def test_ball_query_empty_balls_are_zero(pn, device):
    xyz = _nocs_batch(range(1))
    new_xyz = np.full((1, 9, 3), 5.0, np.float32)  # far from every point
    idx = pn.ball_query(0.1, 16, _dev(xyz, device), _dev(new_xyz, device)).cpu().numpy()
    assert (idx == 0).all()


def test_ball_query_multi_radius_equals_single(device):
    import ctypes
    from captra_amd import _lib
    xyz = _nocs_batch(range(2))
    centres = O.furthest_point_sample(xyz, 512)
    new_xyz = np.take_along_axis(xyz, centres[..., None].astype(np.int64), 1)
    d_xyz, d_new = _dev(xyz, device), _dev(new_xyz, device)
    outs = [torch.full((2, 512, k), -1, dtype=torch.int32, device=device) for _, k in SA_PAIRS]
    radii = (ctypes.c_float * 3)(*[r for r, _ in SA_PAIRS])
    ns = (ctypes.c_int * 3)(*[k for _, k in SA_PAIRS])
    ptrs = (ctypes.c_void_p * 3)(*[o.data_ptr() for o in outs])
    _lib.call("captra_ball_query_multi", 2, 4096, 512, 3, ctypes.cast(radii, ctypes.c_void_p),
              ctypes.cast(ns, ctypes.c_void_p), d_new.data_ptr(), d_xyz.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p))
    for o, (r, k) in zip(outs, SA_PAIRS):
        np.testing.assert_array_equal(o.cpu().numpy(), O.ball_query(r, k, xyz, new_xyz))


# -------------------------------------------------------------------------------- group / gather
@pytest.mark.parametrize("c,n,m,k", [(3, 4096, 512, 32), (6, 4096, 512, 128), (323, 512, 128, 64), (5, 100, 7, 3),
                                       (2, 20000, 16, 8), (40, 512, 128, 128)])
def test_group_points(pn, device, c, n, m, k):
    rng = np.random.default_rng(c + n)
    feat = rng.standard_normal((2, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (2, m, k)).astype(np.int32)
    got = pn.grouping_operation(_dev(feat, device), _dev(idx, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.grouping_operation(feat, idx))


def test_gather_points(pn, device):
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((3, 7, 4096)).astype(np.float32)
    idx = rng.integers(0, 4096, (3, 511)).astype(np.int32)
    got = pn.gather_operation(_dev(feat, device), _dev(idx, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.gather_operation(feat, idx))


def test_group_and_gather_grad(pn, device):
    rng = np.random.default_rng(6)
    feat = torch.from_numpy(rng.standard_normal((2, 5, 300)).astype(np.float32)).to(device).requires_grad_()
    idx = rng.integers(0, 300, (2, 17, 9)).astype(np.int32)
    g = rng.standard_normal((2, 5, 17, 9)).astype(np.float32)
    out = pn.grouping_operation(feat, _dev(idx, device))
    out.backward(_dev(g, device))
    np.testing.assert_allclose(feat.grad.cpu().numpy(), O.grouping_operation_grad(g, idx, 300), rtol=1e-5, atol=1e-5)

    feat2 = torch.from_numpy(rng.standard_normal((2, 5, 300)).astype(np.float32)).to(device).requires_grad_()
    idx2 = rng.integers(0, 300, (2, 40)).astype(np.int32)
    g2 = rng.standard_normal((2, 5, 40)).astype(np.float32)
    pn.gather_operation(feat2, _dev(idx2, device)).backward(_dev(g2, device))
    np.testing.assert_allclose(feat2.grad.cpu().numpy(), O.gather_operation_grad(g2, idx2, 300), rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------- three_nn / interpolate
@pytest.mark.parametrize("n,m", [(512, 128), (4096, 512), (100, 3), (70, 2), (5000, 4500)])
def test_three_nn(pn, device, n, m):
    rng = np.random.default_rng(n + m)
    unknown = (rng.random((2, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    known = (rng.random((2, m, 3), dtype=np.float32) - 0.5).astype(np.float32)
    dist, idx = pn.three_nn(_dev(unknown, device), _dev(known, device))
    d2_ref, idx_ref = O.three_nn(unknown, known)
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(d2_ref), rtol=2e-7, atol=0)  # op layer returns sqrt
    # the kernel itself writes squared distances: exact
    from captra_amd import pointnet2_cuda
    d2 = torch.empty(2, n, 3, device=device)
    ii = torch.empty(2, n, 3, dtype=torch.int32, device=device)
    pointnet2_cuda.three_nn_wrapper(2, n, m, _dev(unknown, device), _dev(known, device), d2, ii)
    np.testing.assert_array_equal(d2.cpu().numpy(), d2_ref)


def test_three_nn_subset_known_has_zero_distance_first(pn, device):
    xyz = _nocs_batch(range(2))
    l1 = O.furthest_point_sample(xyz, 512)
    xyz1 = np.take_along_axis(xyz, l1[..., None].astype(np.int64), 1)
    dist, idx = pn.three_nn(_dev(xyz, device), _dev(xyz1, device))
    d2_ref, idx_ref = O.three_nn(xyz, xyz1)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(d2_ref), rtol=2e-7, atol=0)


@pytest.mark.parametrize("k", [1, 3, 16, 200])
def test_knn(pn, device, k):
    rng = np.random.default_rng(k)
    unknown = (rng.random((2, 300, 3), dtype=np.float32) - 0.5).astype(np.float32)
    known = (rng.random((2, 257, 3), dtype=np.float32) - 0.5).astype(np.float32)
    dist, idx = pn.knn(k, _dev(unknown, device), _dev(known, device))
    d2_ref, idx_ref = O.knn(k, unknown, known)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(d2_ref), rtol=2e-7, atol=0)


@pytest.mark.parametrize("c,m,n", [(256, 128, 512), (128, 512, 4096), (3, 5, 11), (70, 20000, 300)])
def test_three_interpolate(pn, device, c, m, n):
    rng = np.random.default_rng(c + m + n)
    feat = rng.standard_normal((2, c, m)).astype(np.float32)
    idx = rng.integers(0, m, (2, n, 3)).astype(np.int32)
    w = rng.random((2, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    got = pn.three_interpolate(_dev(feat, device), _dev(idx, device), _dev(w, device)).cpu().numpy()
    np.testing.assert_array_equal(got, O.three_interpolate(feat, idx, w))


def test_three_interpolate_grad(pn, device):
    rng = np.random.default_rng(9)
    feat = torch.from_numpy(rng.standard_normal((2, 6, 50)).astype(np.float32)).to(device).requires_grad_()
    idx = rng.integers(0, 50, (2, 80, 3)).astype(np.int32)
    w = rng.random((2, 80, 3), dtype=np.float32)
    g = rng.standard_normal((2, 6, 80)).astype(np.float32)
    pn.three_interpolate(feat, _dev(idx, device), _dev(w, device)).backward(_dev(g, device))
    np.testing.assert_allclose(feat.grad.cpu().numpy(), O.three_interpolate_grad(g, idx, w, 50), rtol=1e-5, atol=1e-5)


# -------------------------------------------------------------------------------------- errors
def test_cpu_tensor_is_rejected_loudly(pn):
    with pytest.raises(RuntimeError):
        pn.furthest_point_sample(torch.zeros(1, 10, 3), 2)


def test_bad_k_raises(pn, device):
    with pytest.raises(RuntimeError):
        pn.knn(201, torch.zeros(1, 4, 3, device=device), torch.zeros(1, 4, 3, device=device))


# ----------------------------------------------------------------------------- degenerate sizes
def test_empty_batches_and_zero_sized_outputs_are_no_ops(device):
    """b = 0, m = 0 or c = 0 through the C ABI: return 0, touch nothing (the reference's launchers would launch a grid of
    zero blocks and print a HIP error)."""
    from captra_amd import _lib
    one = torch.zeros(16, device=device)
    canary = torch.full((16,), 7, dtype=torch.int32, device=device)
    p, q = one.data_ptr(), canary.data_ptr()
    _lib.call("captra_furthest_point_sampling", 0, 10, 4, p, p, q)
    _lib.call("captra_furthest_point_sampling", 2, 10, 0, p, p, q)
    _lib.call("captra_ball_query", 0, 10, 4, 0.1, 4, p, p, q)
    _lib.call("captra_ball_query", 1, 10, 0, 0.1, 4, p, p, q)
    _lib.call("captra_group_points", 0, 3, 10, 4, 4, p, q, p)
    _lib.call("captra_group_points", 1, 0, 10, 4, 4, p, q, p)
    _lib.call("captra_gather_points", 0, 3, 10, 4, p, q, p)
    _lib.call("captra_fps_gather", 0, 10, 4, p, q, p, p)
    torch.cuda.synchronize()
    assert (canary == 7).all() and (one == 0).all()


def test_query_and_group_module(pn, device):
    """pointnet2_utils.QueryAndGroup / GroupAll (reference pointnet2_utils.py:274-330): ball query + grouping + centre
    subtraction + concatenation, features first and relative xyz last."""
    rng = np.random.default_rng(8)
    xyz = (rng.random((2, 300, 3), dtype=np.float32) - 0.5).astype(np.float32)
    new_xyz = xyz[:, :40].copy()
    feat = rng.standard_normal((2, 5, 300)).astype(np.float32)
    idx = O.ball_query(0.2, 16, xyz, new_xyz)
    gx = O.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    gf = O.grouping_operation(feat, idx)
    qg = pn.QueryAndGroup(0.2, 16, use_xyz=True)
    np.testing.assert_array_equal(qg(_dev(xyz, device), _dev(new_xyz, device), _dev(feat, device)).cpu().numpy(),
                                  np.concatenate([gf, gx], 1))
    np.testing.assert_array_equal(qg(_dev(xyz, device), _dev(new_xyz, device)).cpu().numpy(), gx)
    np.testing.assert_array_equal(pn.QueryAndGroup(0.2, 16, use_xyz=False)(_dev(xyz, device), _dev(new_xyz, device),
                                                                           _dev(feat, device)).cpu().numpy(), gf)
    ga = pn.GroupAll(use_xyz=True)(_dev(xyz, device), None, _dev(feat, device)).cpu().numpy()
    np.testing.assert_array_equal(ga, np.concatenate([xyz.transpose(0, 2, 1), feat], 1)[:, :, None])


def test_knn_and_group_module(pn, device):
    """pointnet2_utils.KNNAndGroup (reference pointnet2_utils.py:335-386): k-NN lists (or the caller's idx) + grouping +
    centre subtraction; relative xyz FIRST, features last."""
    rng = np.random.default_rng(18)
    xyz = (rng.random((2, 300, 3), dtype=np.float32) - 0.5).astype(np.float32)
    new_xyz = xyz[:, :40].copy()
    feat = rng.standard_normal((2, 5, 300)).astype(np.float32)
    _, idx = O.knn(16, new_xyz, xyz)
    gx = O.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    gf = O.grouping_operation(feat, idx)
    kg = pn.KNNAndGroup(0.2, 16, use_xyz=True)
    np.testing.assert_array_equal(kg(_dev(xyz, device), _dev(new_xyz, device), None, _dev(feat, device)).cpu().numpy(),
                                  np.concatenate([gx, gf], 1))
    np.testing.assert_array_equal(kg(_dev(xyz, device), _dev(new_xyz, device), _dev(idx, device), _dev(feat, device)).cpu().numpy(),
                                  np.concatenate([gx, gf], 1))
    np.testing.assert_array_equal(kg(_dev(xyz, device), _dev(new_xyz, device)).cpu().numpy(), gx)
    np.testing.assert_array_equal(pn.KNNAndGroup(0.2, 16, use_xyz=False)(_dev(xyz, device), _dev(new_xyz, device), None,
                                                                         _dev(feat, device)).cpu().numpy(), gf)
    # new_xyz defaults to xyz (every point its own centre)
    _, idx_self = O.knn(4, xyz, xyz)
    gself = O.grouping_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx_self) - xyz.transpose(0, 2, 1)[..., None]
    np.testing.assert_array_equal(pn.KNNAndGroup(0.2, 4)(_dev(xyz, device)).cpu().numpy(), gself)


# --------------------------------------------------- backward operators at the shapes of the training step (CSR path)
@pytest.mark.parametrize("c,n,m,k", [(320, 512, 128, 128), (320, 512, 128, 64), (8, 4096, 512, 32), (131, 700, 33, 16)])
def test_group_points_grad_csr_path_vs_float64(pn, device, c, n, m, k):
    """group_points backward where many channels share the index list (C >= 8: the caller-scratch CSR path of
    captra_group_points_grad_ws): every source point sums its list in ascending position order, which is the order of the
    oracle's serial loop -> the SAME bits; and within 1e-6 (relative to the row's scale) of a float64 scatter."""
    from captra_amd import _lib
    rng = np.random.default_rng(c + k)
    idx = np.stack([clouds_ball_idx(rng, n, m, k) for _ in range(2)])
    g = rng.standard_normal((2, c, m, k)).astype(np.float32)
    assert int(_lib.lib().captra_group_points_grad_ws_bytes(2, c, n, m, k)) == 2 * (n + 1 + m * k) * 4
    feat = torch.zeros(2, c, n, device=device, requires_grad=True)
    pn.grouping_operation(feat, _dev(idx, device)).backward(_dev(g, device))
    got = feat.grad.cpu().numpy()
    np.testing.assert_array_equal(got, O.grouping_operation_grad(g, idx, n))
    ref64 = np.zeros((2, c, n), np.float64)
    for b in range(2):
        np.add.at(ref64[b].T, idx[b].reshape(-1), g[b].reshape(c, -1).T.astype(np.float64))
    scale = np.abs(ref64).max(axis=-1, keepdims=True) + 1e-30
    # fp32 sums of up to ~100 terms in a fixed order: a few 1e-7 of the row's scale per term (measured 1.6e-6 at K = 128)
    assert float((np.abs(got - ref64) / scale).max()) <= 4e-6
    # the reference-signature entry (no scratch argument): float atomics, same sums to rounding
    grad_atomic = torch.zeros(2, c, n, device=device)
    g_d, idx_d = _dev(g, device), _dev(idx, device)          # (named: the launch is asynchronous, the operands must outlive it)
    _lib.call("captra_group_points_grad", 2, c, n, m, k, g_d.data_ptr(), idx_d.data_ptr(), grad_atomic.data_ptr())
    torch.cuda.synchronize()
    assert float((np.abs(grad_atomic.cpu().numpy() - ref64) / scale).max()) <= 8e-6


def clouds_ball_idx(rng, n, m, k):
    """Index lists shaped like a ball query's: a few distinct neighbours per centre, padded with the first."""
    idx = np.empty((m, k), np.int32)
    for j in range(m):
        cnt = int(rng.integers(1, k + 1))
        hits = np.sort(rng.choice(n, size=min(cnt, n), replace=False)).astype(np.int32)
        idx[j, :len(hits)] = hits
        idx[j, len(hits):] = hits[0]
    return idx


@pytest.mark.parametrize("c,m,n", [(128, 512, 4096), (256, 128, 512), (9, 50, 333)])
def test_three_interpolate_grad_csr_path_vs_float64(pn, device, c, m, n):
    """three_interpolate backward at FP1 / FP2's shapes (captra_three_interpolate_grad_ws): per-known-point sums in
    ascending (n, j) order = the oracle's loop order -> the same bits; within a few 1e-6 (of the row's scale) of float64."""
    rng = np.random.default_rng(c + m)
    unknown = (rng.random((2, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    known = (rng.random((2, m, 3), dtype=np.float32) - 0.5).astype(np.float32)
    d2, idx = O.three_nn(unknown, known)
    recip = 1.0 / (np.sqrt(d2) + 1e-8)
    w = (recip / recip.sum(-1, keepdims=True)).astype(np.float32)
    g = rng.standard_normal((2, c, n)).astype(np.float32)
    feat = torch.zeros(2, c, m, device=device, requires_grad=True)
    pn.three_interpolate(feat, _dev(idx, device), _dev(w, device)).backward(_dev(g, device))
    got = feat.grad.cpu().numpy()
    np.testing.assert_array_equal(got, O.three_interpolate_grad(g, idx, w, m))
    ref64 = np.zeros((2, c, m), np.float64)
    for b in range(2):
        for j in range(3):
            np.add.at(ref64[b].T, idx[b, :, j], (g[b].astype(np.float64) * w[b, :, j].astype(np.float64)).T)
    scale = np.abs(ref64).max(axis=-1, keepdims=True) + 1e-30
    assert float((np.abs(got - ref64) / scale).max()) <= 4e-6


@pytest.mark.gpu
def test_group_points_multi_equals_single_jobs(device):
    """captra_group_points_multi: all jobs of a level in one launch == the drop-in op job by job (bit-exact copies), at the
    tracking shapes (SA1: 3 radii x 3-channel tensors; SA2: 2 radii x {3, 320} channels) and through the fallback (an odd-sized job)."""
    from captra_amd import fused
    from captra_amd import pointnet2_cuda as pc
    rng = np.random.default_rng(11)
    for n, m, ks, chans in ((4096, 512, (32, 64, 128), (3, 3, 3)), (512, 128, (64, 128), (3, 320, 3, 320)), (300, 7, (5,), (3, 2))):
        B = 3
        feats = {c: torch.from_numpy(rng.standard_normal((B, c, n)).astype(np.float32)).to(device) for c in set(chans)}
        pts, idxs = [], []
        for k in ks:
            idx = torch.from_numpy(rng.integers(0, n, (B, m, k)).astype(np.int32)).to(device)
            for c in chans:
                pts.append(feats[c])
                idxs.append(idx)
        outs = fused.group_points_multi(pts, idxs)
        for p, i, o in zip(pts, idxs, outs):
            ref = torch.empty_like(o)
            pc.group_points_wrapper(B, p.shape[1], n, i.shape[1], i.shape[2], p, i, ref)
            assert torch.equal(o, ref)
            exp = torch.gather(p.unsqueeze(2).expand(-1, -1, i.shape[1], -1), 3, i.long().unsqueeze(1).expand(-1, p.shape[1], -1, -1))
            assert torch.equal(o, exp)


@pytest.mark.gpu
def test_gather_points_grad_takes_the_csr_path_bit_reproducibly(device):
    """GatherOperation.backward with c >= 8 takes the caller-scratch CSR path (captra_gather_points_grad_ws through
    pointnet2_cuda.gather_points_grad_wrapper): equal to a float64 scatter to fp32 rounding, and bit-identical from run to run
    (the float-atomic path is neither ordered nor reproducible)."""
    from captra_amd import _lib as L
    from captra_amd import pointnet2_cuda as pc
    rng = np.random.default_rng(5)
    B, C, N, M = 4, 64, 512, 4096
    assert L.lib().captra_gather_points_grad_ws_bytes(B, C, N, M) > 0
    grad_out = torch.from_numpy(rng.standard_normal((B, C, M)).astype(np.float32)).to(device)
    idx = torch.from_numpy(rng.integers(0, N, (B, M)).astype(np.int32)).to(device)
    outs = []
    for _ in range(3):
        g = torch.zeros(B, C, N, device=device)
        pc.gather_points_grad_wrapper(B, C, N, M, grad_out, idx, g)
        outs.append(g.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.zeros(B, C, N, dtype=torch.float64)
    ref.scatter_add_(2, idx.cpu().long().unsqueeze(1).expand(-1, C, -1), grad_out.cpu().double())
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), atol=4e-6 * float(ref.abs().max()), rtol=0)


@pytest.mark.parametrize("B,N,M,r,K,C,use_xyz", [(2, 4096, 512, 0.05, 32, 0, True), (2, 4096, 512, 0.1, 64, 3, True), (1, 4096, 512, 0.2, 128, 3, True),
                                                  (2, 512, 128, 0.2, 64, 320, True), (2, 512, 128, 0.4, 128, 320, True), (3, 300, 45, 0.2, 16, 5, True),
                                                  (2, 700, 33, 0.3, 20, 7, False), (1, 6000, 100, 0.1, 32, 2, True)])
def test_query_and_group_one_launch_equals_the_two_ops_and_the_oracle(device, B, N, M, r, K, C, use_xyz):
    """captra_query_and_group (the reference's QueryAndGroup module, pointnet2_utils.py:274-310, as ONE launch) against
    captra_ball_query + captra_group_points + the torch centre subtraction / concat, and against the oracle's ball query: the lists
    and every grouped value bit for bit, on the workload's SA1 / SA2 shapes and on ragged ones (centres not a multiple of the
    workgroup's block, K = 16 / 20, clouds beyond one 4096-point tile)."""
    from captra_amd import fused
    from captra_amd import synthetic as clouds
    from captra_amd.pointnet_lib import pointnet2_utils as pn
    rng = np.random.default_rng(B * 1000 + N + K)
    if N == 4096:
        xyz = np.stack([clouds.s_nocs(70 + i)[0] for i in range(B)]).astype(np.float32)
    else:
        xyz = (rng.random((B, N, 3), dtype=np.float32) - 0.5).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, ::max(N // M, 1)][:, :M])
    feat = rng.standard_normal((B, C, N)).astype(np.float32) if C else None
    x, nx = _dev(xyz, device), _dev(new_xyz, device)
    f = _dev(feat, device) if C else None
    idx = pn.ball_query(r, K, x, nx)
    np.testing.assert_array_equal(idx.cpu().numpy(), O.ball_query(r, K, xyz, new_xyz))
    gx = pn.grouping_operation(x.transpose(1, 2).contiguous(), idx) - nx.transpose(1, 2).unsqueeze(-1)
    want = gx if f is None else (torch.cat([pn.grouping_operation(f, idx), gx], dim=1) if use_xyz else pn.grouping_operation(f, idx))
    got, got_idx = fused.query_and_group(r, K, x, nx, f, use_xyz, want_idx=True)
    torch.cuda.synchronize()
    assert torch.equal(got_idx, idx)
    assert torch.equal(got, want)
    assert torch.equal(fused.query_and_group(r, K, x, nx, f, use_xyz), want)       # (no list output)
    # the module itself (inference: the one launch; a K the kernel does not take falls back to the ops)
    assert torch.equal(pn.QueryAndGroup(r, K, use_xyz)(x, nx, f), want)
    assert fused.query_and_group(r, 18, x, nx, f, use_xyz) is None
