"""Property tests of the CPU oracle (hypothesis): the invariants the reference's algorithms guarantee by construction, on
random small clouds -- independent of the golden vectors, so that an oracle bug cannot hide behind a matching fixture."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import ops as O

clouds_st = st.integers(0, 2 ** 31 - 1).flatmap(lambda seed: st.tuples(st.just(seed), st.integers(5, 200), st.integers(1, 40)))


def _cloud(seed, n, dup=False):
    rng = np.random.default_rng(seed)
    pts = (rng.random((1, n, 3), dtype=np.float32) - 0.5)
    if dup and n > 4:
        pts[0, n // 2:] = pts[0, : n - n // 2]          # exact duplicates
    return pts


@settings(max_examples=60, deadline=None)
@given(clouds_st, st.booleans())
def test_fps_invariants(args, dup):
    seed, n, m = args
    m = min(m, n)
    xyz = _cloud(seed, n, dup)
    idx = O.furthest_point_sample(xyz, m)[0]
    assert idx[0] == 0 and idx.min() >= 0 and idx.max() < n
    # greedy property: pick j maximises the distance to the picks before it; ties go to the lowest index
    d = np.full(n, 1e10, np.float32)
    for j in range(1, m):
        p = xyz[0, idx[j - 1]]
        diff = xyz[0] - p
        d2 = ((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).astype(np.float32)
        d = np.minimum(d, d2)
        assert idx[j] == int(np.argmax(d)), (j, idx[j], int(np.argmax(d)))     # np.argmax returns the first maximum
    if not dup:
        assert len(set(idx.tolist())) == m


@settings(max_examples=60, deadline=None)
@given(clouds_st, st.floats(0.05, 0.6), st.sampled_from([4, 16, 32]))
def test_ball_query_invariants(args, radius, k):
    seed, n, m = args
    m = min(m, n)
    xyz = _cloud(seed, n)
    new_xyz = xyz[:, :m].copy()
    idx = O.ball_query(np.float32(radius), k, xyz, new_xyz)[0]
    r2 = np.float32(radius) * np.float32(radius)
    for c in range(m):
        diff = xyz[0] - new_xyz[0, c]
        d2 = ((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).astype(np.float32)
        inside = np.nonzero(d2 < r2)[0]
        want = inside[:k]
        assert len(want) >= 1                                   # the centre itself (distance 0) is always inside
        np.testing.assert_array_equal(idx[c, :len(want)], want)  # first K in index order
        assert (idx[c, len(want):] == want[0]).all()             # padded with the first hit


@settings(max_examples=40, deadline=None)
@given(clouds_st)
def test_three_nn_invariants(args):
    seed, n, m = args
    m = max(3, min(m, n))
    unknown = _cloud(seed, n)
    known = _cloud(seed + 1, m)
    dist2, idx = O.three_nn(unknown, known)                      # SQUARED distances, as the kernel writes them
    assert (np.diff(dist2[0], axis=1) >= 0).all()               # ascending
    diff = unknown[0, :, None, :] - known[0, None, :, :]
    d2 = ((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]).astype(np.float32)
    np.testing.assert_array_equal(dist2[0], np.sort(d2, axis=1)[:, :3])
    np.testing.assert_array_equal(np.take_along_axis(d2, idx[0].astype(np.int64), axis=1), dist2[0])


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 40), st.integers(1, 50), st.integers(1, 300))
def test_pointwise_mlp_is_the_fmaf_chain(seed, cin, cout, length):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((1, cin, length)).astype(np.float32)
    w = rng.standard_normal((cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = O.pointwise_mlp(x, w, b, 1)[0]
    # the same chain in extended precision stays within one rounding per step of the fp32 chain
    ref = np.maximum(b[:, None].astype(np.float64) + w.T.astype(np.float64) @ x[0].astype(np.float64), 0)
    bound = 4e-7 * (np.abs(b)[:, None] + np.abs(w.T) @ np.abs(x[0])) * max(cin, 1) ** 0.5 + 1e-30
    assert (np.abs(y - ref) <= bound).all()
    assert (y >= 0).all()
