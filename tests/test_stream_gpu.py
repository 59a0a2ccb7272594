"""Streamed level-1 sampling (captra_fps_gather_part + captra_launch_opts::centre_m0 / centre_mc): the sampler cut into parts and the ball query /
small-input SA scales run per window of centres must produce what the one-launch forms produce, bit for bit -- furthest-point
sampling is one loop (reference sampling_gpu.cu:93-209), a centre's neighbour list and pooled features depend on that centre
only (ball_query_gpu.cu:9-45, pointnet_utils.py:228-248)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("B,N,M,cuts", [(3, 4096, 512, (128, 256, 384)), (2, 4096, 512, (256,)), (1, 4096, 512, (8, 9, 500)),
                                         (5, 512, 128, (64,)), (2, 1000, 77, (1, 40)), (4, 2048, 256, (64, 128, 192))])
def test_fps_parts_equal_one_launch(device, B, N, M, cuts):
    from captra_amd import fused
    from captra_amd import synthetic as clouds
    if N == 4096:
        xyz = np.stack([clouds.s_nocs(50 + i)[0] for i in range(B)]).astype(np.float32)
        xyz[0, 100:200] = xyz[0, 0:100]                           # duplicated points: ties
    else:
        xyz = (np.random.default_rng(N + M).random((B, N, 3), dtype=np.float32) - 0.5)
    x = _dev(xyz, device)
    idx, n3, cn = fused.fps_gather(x, M)
    bufs = fused.fps_gather_parts(x, M)
    for t in bufs[:3]:
        t.fill_(-7)
    edges = (0,) + tuple(cuts) + (M,)
    for j0, j1 in zip(edges[:-1], edges[1:]):
        fused.fps_gather_part(x, M, j0, j1, bufs)
        torch.cuda.synchronize()
        assert torch.equal(bufs[0][:, :j1], idx[:, :j1]) and (bufs[0][:, j1:] == -7).all(), (j0, j1)
    assert torch.equal(bufs[0], idx) and torch.equal(bufs[1], n3) and torch.equal(bufs[2], cn)


@pytest.mark.parametrize("B,N,M,step", [(3, 4096, 512, 128), (2, 512, 128, 32), (2, 700, 45, 16)])
def test_ball_query_windows_equal_one_launch(device, B, N, M, step):
    from captra_amd import fused
    rng = np.random.default_rng(N + M)
    xyz = _dev(rng.random((B, N, 3), dtype=np.float32) - 0.5, device)
    new = xyz[:, :M].contiguous()
    radii, ks = (0.05, 0.1, 0.2), (32, 64, 128)
    want = fused.ball_query_multi(radii, ks, xyz, new)
    got = [torch.full_like(w, -3) for w in want]
    for m0 in range(0, M, step):
        mc = min(step, M - m0)
        with fused.centre_window(m0, mc):
            fused.ball_query_multi(radii, ks, xyz, new, outs=got)
        for g, w in zip(got, want):
            assert torch.equal(g[:, :m0 + mc], w[:, :m0 + mc]) and (g[:, m0 + mc:] == -3).all()
    # no window left behind
    again = fused.ball_query_multi(radii, ks, xyz, new)
    assert all(torch.equal(a, w) for a, w in zip(again, want))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cfeat,chans,n,m,k,B,step", [(0, (64, 96, 128), 4096, 512, 128, 3, 128), (3, (64, 64, 128), 4096, 512, 64, 2, 256),
                                                       (3, (32, 32, 64), 4096, 512, 32, 2, 128), (0, (64, 96, 128), 4096, 512, 128, 1, 128),
                                                       (3, (64, 96, 128), 900, 40, 128, 2, 16)])
def test_sa_scale_windows_equal_one_launch(device, dtype, cfeat, chans, n, m, k, B, step):
    from captra_amd import fused
    rng = np.random.default_rng(cfeat + sum(chans) + k + B)
    xyz_cn = _dev(rng.random((B, 3, n), dtype=np.float32) - 0.5, device)
    feat = _dev(rng.standard_normal((B, cfeat, n)).astype(np.float32), device) if cfeat else None
    new_xyz = _dev(rng.random((B, m, 3), dtype=np.float32) - 0.5, device)
    idx = _dev(rng.integers(0, n, (B, m, k)).astype(np.int32), device)
    dims = (cfeat + 3,) + chans
    packed = [fused.pack(_dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device),
                         _dev(rng.standard_normal(dims[i + 1]).astype(np.float32), device)) for i in range(3)]

    def run(out):
        if dtype == "bf16":
            fused.sa_scale_bf16(feat, xyz_cn, new_xyz, idx, packed, out, 4)
        else:
            fused.sa_scale_fused(feat, xyz_cn, new_xyz, idx, packed, out, 4)

    fused.set_mlp_dtype(dtype)
    try:
        want = torch.full((B, chans[2] + 9, m), -1.0, device=device)
        run(want)
        got = torch.full((B, chans[2] + 9, m), -1.0, device=device)
        for m0 in range(0, m, step):
            mc = min(step, m - m0)
            with fused.centre_window(m0, mc):
                run(got)
            assert torch.equal(got[:, :, :m0 + mc], want[:, :, :m0 + mc]), (m0, mc)
            assert (got[:, :, m0 + mc:] == -1).all()
    finally:
        fused.set_mlp_dtype("fp32")
    assert torch.equal(got, want)


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("chunks", [2, 4])
def test_track_step_with_streamed_sampler_equals_plain_step(device, mlp_dtype, chunks):
    """EvalTrackModel with sampler_chunks = 2 / 4: the first level's sampling on a stream of its own in parts, both networks'
    first level walking the centres as they are picked -- every frame's pose equals the plain step's bit for bit, eager and
    as a captured graph (chained over the frames of a trajectory, so a wrong pick anywhere would surface)."""
    from captra_amd.graph import TrackStepGraph
    from tests.test_model_gpu import _trainer
    trainer, cfg, sd, data = _trainer("bottle", device)
    model = trainer.model.eval()
    model.track_cfg["gt_label"] = False
    model.mlp_dtype = mlp_dtype if mlp_dtype != "fp32" else None
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
        model.sampler_chunks = 0
        want = loop(eager)
        model.sampler_chunks = chunks
        got = loop(eager)
        graph = TrackStepGraph(model, model.feed_dict[1]["points"], model.feed_dict[1]["points_mean"], pose0)
        got_g = loop(lambda i, pe: graph.replay(model.feed_dict[i]["points"], model.feed_dict[i]["points_mean"], pe))
    finally:
        model.sampler_chunks = 0
        model.mlp_dtype = None
    for i, (a, b, c) in enumerate(zip(want, got, got_g)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, "eager")
            assert torch.equal(a[k], c[k]), (i, k, "graph")
