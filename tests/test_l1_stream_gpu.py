"""Level-1 stream kernel (captra_sa1_stream_bf16): sampler workgroups publishing their picks while the rest of the chip runs the ball
query and the SA1 scales of the published windows.  Every output must equal what the three separate launches produce -- furthest
point sampling (reference sampling_gpu.cu:93-209), ball query (ball_query_gpu.cu:9-45) and the loop over radii of
PointNetSetAbstractionMsg.forward (pointnet_utils.py:228-248) -- bit for bit, under uneven load and repeated launches (the hand-off
is granules across XCDs: a stale read would show up as a wrong pick somewhere)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WIDTHS = ((32, 32, 64), (64, 64, 128), (64, 96, 128))
KS = (32, 64, 128)
RADII = (0.05, 0.1, 0.2)


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _module(cf, seed, device, npoint=512):
    from captra_amd import fused
    rng = np.random.default_rng(seed)
    folded = []
    for chans in WIDTHS:
        dims = (cf + 3,) + chans
        folded.append([fused.pack(_dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device),
                                  _dev(0.1 * rng.standard_normal(dims[i + 1]).astype(np.float32), device)) for i in range(3)])
    return SimpleNamespace(training=False, knn=False, nsample_list=list(KS), radius_list=list(RADII), npoint=npoint, _folded=folded)


def _clouds(B, N, device, seed=0):
    from captra_amd import synthetic as clouds
    if N == 4096:
        xyz = np.stack([clouds.s_nocs(seed + 50 + i)[0] for i in range(B)]).astype(np.float32)
        xyz[0, 100:200] = xyz[0, 0:100]                           # duplicated points: ties in the sampler
    else:
        xyz = (np.random.default_rng(seed + N).random((B, N, 3), dtype=np.float32) - 0.5) * 0.8
    return _dev(xyz, device)


def _reference(x_n3, x_cn, mods, feats):
    from captra_amd import fused
    M = mods[0].npoint
    idx, n3, cn = fused.fps_gather(x_n3, M)
    lists = fused.ball_query_multi(RADII, KS, x_n3, n3)
    outs = []
    for mod, feat in zip(mods, feats):
        out = torch.zeros(x_n3.shape[0], 320, M, device=x_n3.device)
        off = 0
        for layers, l in zip(mod._folded, lists):
            fused.sa_scale_bf16(feat, x_cn, n3, l, layers, out, off)
            off += layers[-1].cout
        outs.append(out)
    return idx, n3, cn, lists, outs


def _check(got, want, tag=""):
    idx, n3, cn, lists, outs, scratch = got
    from captra_amd import fused
    assert not fused.sa1_stream_gave_up(scratch), f"{tag}: a consumer gave up waiting for the sampler"
    assert torch.equal(idx, want[0]), f"{tag}: picks differ"
    assert torch.equal(n3, want[1]) and torch.equal(cn, want[2]), f"{tag}: sampled coordinates differ"
    for s, (g, w) in enumerate(zip(lists, want[3])):
        assert torch.equal(g, w), f"{tag}: neighbour lists of scale {s} differ"
    for i, (g, w) in enumerate(zip(outs, want[4])):
        assert torch.equal(g, w), f"{tag}: pooled features of network {i} differ (max abs {float((g - w).abs().max())})"


@pytest.mark.parametrize("B,N,M,cfs", [(1, 4096, 512, (0, 3)), (3, 4096, 512, (0, 3)), (32, 4096, 512, (0, 3)), (2, 4096, 512, (3, 0)),
                                       (4, 4096, 512, (0,)), (2, 4096, 512, (3,)), (2, 3000, 256, (0, 3)), (5, 1000, 64, (3,)),
                                       (64, 4096, 512, (0, 3))])
def test_stream_kernel_equals_three_launches(device, B, N, M, cfs):
    from captra_amd import fused
    fused.set_mlp_dtype("bf16")
    try:
        x_n3 = _clouds(B, N, device)
        x_cn = x_n3.transpose(1, 2).contiguous()
        mods = [_module(cf, 11 + i, device, npoint=M) for i, cf in enumerate(cfs)]
        feats = [x_cn.clone() * 0.5 if cf else None for cf in cfs]
        assert fused.sa1_stream_supported(N, mods, cfs)
        want = _reference(x_n3, x_cn, mods, feats)
        planes = fused.bq_planes(x_n3)
        m2 = min(128, M // 2)
        want2 = fused.fps_gather(want[1], m2)
        for rep in range(3):
            got = fused.sa1_stream_bf16(x_n3, x_cn, mods, feats, planes=planes if rep else None, m2=m2 if rep != 1 else 0)
            torch.cuda.synchronize()
            _check(got[:6], want, f"rep {rep}")
            if rep != 1:
                for g, w, what in zip(got[6], want2, ("picks", "coordinates (B,m2,3)", "coordinates (B,3,m2)")):
                    assert torch.equal(g, w), f"rep {rep}: level-2 {what} differ"
    finally:
        fused.set_mlp_dtype("fp32")


def test_stream_kernel_under_uneven_load_and_small_grids(device):
    """Another stream keeps part of the chip busy (so the stream kernel's workgroups are placed unevenly and late) and the grid is
    shrunk to fewer workgroups than tickets need at once: the hand-off must not depend on placement, timing or residency."""
    from captra_amd import _lib as L
    from captra_amd import fused
    fused.set_mlp_dtype("bf16")
    try:
        B = 16
        x_n3 = _clouds(B, 4096, device, seed=7)
        x_cn = x_n3.transpose(1, 2).contiguous()
        mods = [_module(0, 3, device), _module(3, 4, device)]
        feats = [None, x_cn.clone()]
        want = _reference(x_n3, x_cn, mods, feats)
        side = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device=device)
        for grid, prio, fine, whole in ((0, 1, 32, 0), (B + 1, 1, 0, 16), (B + 7, 0, 64, 256), (300, 1, 512, 3), (1024, 1, 32, 256 + 12)):
            L.lib().captra_sa1_stream_set_grid(grid, prio)
            L.lib().captra_sa1_stream_set_fine(fine)
            L.lib().captra_sa1_stream_set_whole(whole)
            for rep in range(4):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a = torch.tanh(a @ a * 1e-3)
                got = fused.sa1_stream_bf16(x_n3, x_cn, mods, feats)
                torch.cuda.synchronize()
                _check(got, want, f"grid {grid} rep {rep}")
    finally:
        L.lib().captra_sa1_stream_set_grid(0, 1)
        L.lib().captra_sa1_stream_set_fine(32)
        L.lib().captra_sa1_stream_set_whole(0)
        fused.set_mlp_dtype("fp32")


@pytest.mark.parametrize("tag", ["bottle", "camera"])
def test_track_step_with_level1_stream_equals_plain_step(device, tag):
    """EvalTrackModel in bf16 mode with both networks' first level inside the sampler's launch (model.l1_stream): every frame's
    pose equals the plain step's (sampler, ball query and SA1 scales as launches of their own) bit for bit, eager and as a captured
    graph, chained over the frames of a trajectory -- and no consumer gave up."""
    from captra_amd.graph import TrackStepGraph
    from tests.test_model_gpu import _trainer
    trainer, cfg, sd, data = _trainer(tag, device)
    model = trainer.model.eval()
    model.track_cfg["gt_label"] = False
    model.mlp_dtype = "bf16"
    model.set_data(data)
    pose0 = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}

    def loop(step):
        pe, out = pose0, []
        for i in range(1, len(data)):
            pe = step(i, pe)
            out.append({k: v.clone() for k, v in pe.items()})
        torch.cuda.synchronize()
        return out

    def eager(i, pe):
        with torch.no_grad():
            return model.track_step(model.feed_dict[i], model.npcs_feed_dict[i], pe)[1]

    try:
        model.l1_stream = False
        want = loop(eager)
        model.l1_stream = True
        model._l1_scratch = None
        got = loop(eager)
        assert model._l1_scratch is not None, "the level-1 stream kernel did not run"
        model.check_l1_stream()
        graph = TrackStepGraph(model, model.feed_dict[1]["points"], model.feed_dict[1]["points_mean"], pose0)
        got_g = loop(lambda i, pe: graph.replay(model.feed_dict[i]["points"], model.feed_dict[i]["points_mean"], pe))
        model.check_l1_stream()
    finally:
        model.l1_stream = True
        model.mlp_dtype = None
    for i, (a, b, c) in enumerate(zip(want, got, got_g)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, "eager")
            assert torch.equal(a[k], c[k]), (i, k, "graph")


@pytest.mark.parametrize("overlap", [True, False])
def test_captured_step_with_level1_stream_at_sixteen_trajectories(device, overlap):
    """The captured step (hipGraph replay) with the level-1 stream kernel at 16 trajectories, the two networks side by side and one
    after the other.  The second form is ONE linear chain of graph nodes: with the kernel's scratch zeroed by a hipMemsetAsync node
    the fill overlapped the kernel from 64 KiB on (granules zeroed after the samplers had published them: consumers gave up, wild
    gathers, memory faults) -- it is zeroed by a kernel now (common.h captra_zero_async); every replayed pose must equal the eager
    step's, and no consumer may give up."""
    import bench
    from captra_amd.graph import TrackStepGraph
    cfg, sd, model, data = bench.build_workload(16, device, category="bottle", traj_seed=0, mlp_dtype="bf16")
    model.overlap_nets = overlap
    pose0 = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
    n = len(model.feed_dict)

    def loop(step, k=5):
        pe, out = pose0, []
        for i in range(1, k):
            pe = step(1 + (i - 1) % (n - 1), pe)
            out.append({kk: v.clone() for kk, v in pe.items()})
            torch.cuda.synchronize()
            model.check_l1_stream()
        return out

    def eager(i, pe):
        with torch.no_grad():
            return model.track_step(model.feed_dict[i], model.npcs_feed_dict[i], pe)[1]

    want = loop(eager)
    assert model._l1_scratch is not None, "the level-1 stream kernel did not run"
    graph = TrackStepGraph(model, model.feed_dict[1]["points"], model.feed_dict[1]["points_mean"], pose0)
    got = loop(lambda i, pe: graph.replay(model.feed_dict[i]["points"], model.feed_dict[i]["points_mean"], pe))
    for i, (a, b) in enumerate(zip(want, got)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)
