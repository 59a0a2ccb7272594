"""GPU parity of the LDS-tiled bf16 dense kernels (csrc/tile_bf16.hip; BASELINE.json configs[2]'s arithmetic).

The per-layer contract is the streaming kernels' (csrc/dense_bf16.hip): y = act(b + sum_k bf16(w) bf16(x)), k ascending, fp32
accumulation -- the same MFMA sequence, so the tiled layer must reproduce the streaming layer BIT FOR BIT, with and without the
on-load GroupNorm; its statistics are the partial sums of the fp32 outputs.  The fused head (layers 1 + 2 in one launch, y1
never stored) is checked against the float64 evaluation of the same chain on bf16-rounded operands and against the torch
Sequential (reference network/models/blocks.py:147-165).
"""
import numpy as np
import pytest
import torch

from tests.test_model_gpu import _bf16_layer, _bf16_round, _dense_to_pm, _dev, _pm_to_dense

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,l", [(512, 512, 4096), (512, 256, 1000), (128, 512, 333), (256, 128, 512), (576, 256, 512),
                                         (512, 1024, 128), (40, 70, 77), (1536, 256, 128), (200, 64, 129)])
def test_tile_layer_equals_streaming_layer_bit_for_bit(device, cin, cout, l):
    from captra_amd import fused
    rng = np.random.default_rng(cin * 3 + cout + l)
    B = 2
    x = _bf16_round(rng.standard_normal((B, cin, l)).astype(np.float32))
    w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lin = fused.pack(_dev(w, device), _dev(b, device))
    xpm = _dense_to_pm(x).to(device)
    assert fused.dense_bf16_tile_supported(xpm, lin)
    for act in (fused.ACT_NONE, fused.ACT_RELU):
        want = fused.pointwise_mlp_bf16pm(xpm, lin, l, in_pm=True, out_pm=True, act=act)
        got = fused.dense_bf16_tile(xpm, lin, act=act)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # against the float64 contract
    ref = _bf16_layer(x, w, b, 0)
    got = _pm_to_dense(fused.dense_bf16_tile(xpm, lin), cout)
    assert np.abs(got - _bf16_round(ref)).max() <= 2.0 ** -7 * np.abs(ref).max()
    assert np.mean(got == _bf16_round(ref)) > 0.995
    # on-load GroupNorm: x -> bf16(relu(a x + b)), bit-identical to the streaming kernels' transform
    ab = rng.standard_normal((B, cin, 2)).astype(np.float32)
    want = fused.pointwise_mlp_bf16pm(xpm, lin, l, in_pm=True, out_pm=True, ab=_dev(ab, device))
    got, st = fused.dense_bf16_tile(xpm, lin, ab=_dev(ab, device), with_stats=True)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # statistics: partial sums of the fp32 outputs per chunk of 128 positions; the stored tensor is those outputs rounded
    xn = _bf16_round(np.maximum(np.float32(ab[:, :, 0:1]) * x + np.float32(ab[:, :, 1:2]), 0).astype(np.float32))
    y64 = np.einsum("kc,bkl->bcl", _bf16_round(w).astype(np.float64), xn.astype(np.float64)) + b[None, :, None]
    st = st.cpu().numpy().astype(np.float64)
    T = (l + 127) // 128
    assert st.shape == (B, T, cout, 2)
    scale = max(1.0, float(np.abs(y64).max()))
    for t in range(T):
        seg = y64[:, :, 128 * t:128 * t + 128]
        np.testing.assert_allclose(st[:, t, :, 0], seg.sum(-1), atol=2e-2 * scale, rtol=1e-3)      # (a rounding tie in the transform flips an operand)
        np.testing.assert_allclose(st[:, t, :, 1], (seg ** 2).sum(-1), atol=2e-2 * scale * scale, rtol=2e-3)
    st2 = fused.dense_bf16_tile(xpm, lin, ab=_dev(ab, device), with_stats=True)[1]
    assert torch.equal(st2.cpu(), torch.from_numpy(st.astype(np.float32)))                      # no atomics: run-to-run identical
    # a bias per cloud
    if cout % 32 == 0:
        bias_bc = rng.standard_normal((B, cout)).astype(np.float32)
        got = _pm_to_dense(fused.dense_bf16_tile(xpm, lin, bias_bc=_dev(bias_bc, device)), cout)
        refb = np.einsum("kc,bkl->bcl", _bf16_round(w).astype(np.float64), x.astype(np.float64)) + bias_bc[:, :, None]
        assert np.abs(got - _bf16_round(refb.astype(np.float32))).max() <= 2.0 ** -7 * np.abs(refb).max()


def _gn_coeffs(y, gamma, beta, eps, cpg):
    """GroupNorm(y) = a y + b per (cloud, channel), float64."""
    B, C, L = y.shape
    g = y.reshape(B, C // cpg, cpg * L)
    mean, var = g.mean(-1), g.var(-1)
    rstd = 1.0 / np.sqrt(var + eps)
    a = gamma[None, :] * np.repeat(rstd, cpg, axis=1)
    return a, beta[None, :] - np.repeat(mean, cpg, axis=1) * a


@pytest.mark.parametrize("cin,l", [(128, 4096), (128, 1000), (96, 77)])
def test_head12_one_launch_vs_float64_chain(device, cin, l):
    """captra_head12_bf16: statistics pass + fused layers 1-2 against y2 = W2 bf16(relu(GN(W1 x + b1))) + b2 in float64."""
    from captra_amd import fused
    rng = np.random.default_rng(cin + l)
    B, cpg, eps = 2, 2, 1e-5
    x = _bf16_round(rng.standard_normal((B, cin, l)).astype(np.float32))
    w1 = (rng.standard_normal((cin, 512)) / np.sqrt(cin)).astype(np.float32)
    b1 = rng.standard_normal(512).astype(np.float32)
    w2 = (rng.standard_normal((512, 512)) / np.sqrt(512)).astype(np.float32)
    b2 = rng.standard_normal(512).astype(np.float32)
    g1, be1 = rng.uniform(0.5, 1.5, 512), rng.uniform(-0.3, 0.3, 512)
    lin1, lin2 = fused.pack(_dev(w1, device), _dev(b1, device)), fused.pack(_dev(w2, device), _dev(b2, device))
    xpm = _dense_to_pm(x).to(device)
    assert fused.head12_bf16_supported(xpm, lin1, lin2)
    st1 = fused.head12_bf16_stats(xpm, lin1)
    y1 = np.einsum("kc,bkl->bcl", _bf16_round(w1).astype(np.float64), x.astype(np.float64)) + b1[None, :, None]
    s = st1.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(s[..., 0].sum(1), y1.sum(-1), atol=1e-3 * np.sqrt(l) * max(1.0, np.abs(y1).max()), rtol=1e-5)
    np.testing.assert_allclose(s[..., 1].sum(1), (y1 ** 2).sum(-1), rtol=2e-5)
    gam, bet = _dev(g1.astype(np.float32), device), _dev(be1.astype(np.float32), device)
    ab1 = fused.gn_finalize(st1, 512 // cpg, gam, bet, eps, l, tile_major=True)
    a64, b64 = _gn_coeffs(y1, g1.astype(np.float32).astype(np.float64), be1.astype(np.float32).astype(np.float64), eps, cpg)
    ab = ab1.cpu().numpy()
    np.testing.assert_allclose(ab[..., 0], a64, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ab[..., 1], b64, rtol=1e-3, atol=1e-4)
    y2pm, st2 = fused.head12_bf16(xpm, lin1, ab1, lin2)
    h = _bf16_round(np.maximum(ab[..., 0:1].astype(np.float64) * y1 + ab[..., 1:2], 0).astype(np.float32))
    y2 = np.einsum("kc,bkl->bcl", _bf16_round(w2).astype(np.float64), h.astype(np.float64)) + b2[None, :, None]
    got = _pm_to_dense(y2pm, 512)
    scale = float(np.abs(y2).max())
    err = np.abs(got - y2)
    assert err.max() <= 3e-2 * scale and err.mean() <= 3e-3 * scale, (err.max() / scale, err.mean() / scale)
    s2 = st2.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(s2[..., 0].sum(1), y2.sum(-1), atol=3e-2 * l ** 0.5 * scale, rtol=1e-3)
    np.testing.assert_allclose(s2[..., 1].sum(1), (y2 ** 2).sum(-1), rtol=5e-3)
    # run-to-run identical (no atomics), also for the stored tensor
    y2b, st2b = fused.head12_bf16(xpm, lin1, ab1, lin2)
    assert torch.equal(y2b.view(torch.int16), y2pm.view(torch.int16)) and torch.equal(st2b, st2)


def test_rotation_head_chain_tiled_vs_streaming_and_torch(device):
    """MLPConv1d(128 -> 512 -> 512 -> 256 -> 3, GroupNorm) fed point-major in the bf16 mode: the round-4 route (fused layers 1-2,
    tiled layer 3) against the round-3 route (streaming kernels) and the torch fp32 Sequential."""
    from captra_amd import fused
    from captra_amd.blocks import MLPConv1d
    torch.manual_seed(11)
    head = MLPConv1d(128, [512, 512, 256, 3], bn=True, gn=True, last_activation="none").to(device).eval()
    with torch.no_grad():
        for m in head.model:
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    x = torch.randn(3, 128, 1000, device=device)
    xpm = _dense_to_pm(_bf16_round(x.cpu().numpy())).to(device)
    with torch.no_grad():
        ref = head.model(torch.from_numpy(_bf16_round(x.cpu().numpy())).to(device))
        with fused.use_mlp_dtype("bf16"):
            got = head(fused.PMTensor(xpm, 128))
            fused.USE_TILE_BF16 = False
            try:
                old = head(fused.PMTensor(xpm, 128))
            finally:
                fused.USE_TILE_BF16 = True
    assert got.shape == ref.shape and got.dtype == torch.float32
    scale = float(ref.abs().max())
    for name, other, mx, mean in (("torch", ref, 5e-2, 6e-3), ("round-3 route", old, 5e-2, 6e-3)):
        d = (got - other).abs()
        assert float(d.max()) <= mx * scale and float(d.mean()) <= mean * scale, (name, float(d.max()) / scale, float(d.mean()) / scale)


@pytest.mark.parametrize("cin,csplit,cout,l", [(515, 3, 256, 128), (576, 576, 256, 512), (512, 512, 256, 128), (134, 6, 128, 4096),
                                               (40, 17, 70, 76), (320, 320, 32, 512), (512, 100, 1024, 128)])
def test_tile_layer_channel_major_input_and_fp32_outputs(device, cin, csplit, cout, l):
    """captra_dense_bf16_tile_ex with a channel-major fp32 input (optionally two tensors standing for their concat) and its three
    output forms, against the streaming kernel on the concatenated input and the float64 contract; the pooled form is the exact
    max of the fp32 form."""
    from captra_amd import fused
    rng = np.random.default_rng(cin + 5 * cout + l)
    B = 3
    x = rng.standard_normal((B, cin, l)).astype(np.float32)
    w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lin = fused.pack(_dev(w, device), _dev(b, device))
    xd = _dev(x, device)
    xa = xd[:, :csplit].contiguous()
    xb = xd[:, csplit:].contiguous() if csplit < cin else None
    for act in (fused.ACT_NONE, fused.ACT_RELU):
        want_pm = fused.pointwise_mlp_bf16pm(xd, lin, l, in_pm=False, out_pm=True, act=act)
        want_cm = fused.pointwise_mlp_bf16pm(xd, lin, l, in_pm=False, out_pm=False, act=act)
        got_pm = fused.dense_bf16_tile(xa, lin, act=act, x2=xb, out_mode=fused.OUT_PM)
        got_cm = fused.dense_bf16_tile(xa, lin, act=act, x2=xb, out_mode=fused.OUT_CM)
        # (the streaming kernel multiplies a k-step's 16 channels in natural order, this one in slot order: the MFMA's internal
        # summation differs, so fp32 results agree to rounding and a bf16 output may flip by one ulp on a tie)
        scale = max(1.0, float(want_cm.abs().max()))
        assert float((got_cm - want_cm).abs().max()) <= 2e-5 * scale
        a, w_ = got_pm.float(), want_pm.float()
        assert float((a - w_).abs().max()) <= 2.0 ** -7 * scale and float((a == w_).float().mean()) > 0.995
        if l <= 128:
            got_mx = fused.dense_bf16_tile(xa, lin, act=act, x2=xb, out_mode=fused.OUT_MAX)
            assert torch.equal(got_mx, got_cm.max(dim=2, keepdim=True)[0])
    ref = _bf16_layer(x, w, b, 1)
    np.testing.assert_allclose(got_cm.cpu().numpy(), ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())), rtol=0)
    # point-major in -> fp32 out / pooled
    if cout >= 64:
        xpm = _dense_to_pm(_bf16_round(x)).to(device)
        want = fused.pointwise_mlp_bf16pm(xpm, lin, l, in_pm=True, out_pm=False, act=fused.ACT_RELU)
        assert torch.equal(fused.dense_bf16_tile(xpm, lin, act=fused.ACT_RELU, out_mode=fused.OUT_CM), want)
        if l <= 128:
            assert torch.equal(fused.dense_bf16_tile(xpm, lin, act=fused.ACT_RELU, out_mode=fused.OUT_MAX), want.max(dim=2, keepdim=True)[0])


def test_gemv_bf16_vs_float64(device):
    from captra_amd import fused
    rng = np.random.default_rng(4)
    for B, cin, cout in [(32, 1024, 256), (3, 77, 70)]:
        v = rng.standard_normal((B, cin)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        got = fused.gemv_bf16(_dev(v, device), fused.pack(_dev(w, device), _dev(b, device))).cpu().numpy()
        want = _bf16_round(v).astype(np.float64) @ _bf16_round(w).astype(np.float64) + b
        np.testing.assert_allclose(got, want, atol=2e-5 * max(1.0, np.abs(want).max()), rtol=0)


def test_sa3_and_fp_levels_tiled_vs_streaming_route(device):
    """The 128- / 512-point levels in the bf16 mode (pointnet_utils.py:253-343: group_all SA3, FP3 with one source vector per cloud,
    FP2): the round-4 route (LDS-tiled kernels, no concat, pooled epilogue, per-cloud product as a GEMV) against the round-3 route
    (streaming kernels) -- the same contract layer by layer, so agreement to accumulation-order noise."""
    from captra_amd import fused
    from captra_amd.pointnet_utils import PointNetFeaturePropagation, PointNetSetAbstraction
    torch.manual_seed(2)
    B = 4
    sa3 = PointNetSetAbstraction(None, None, None, 512 + 3, [256, 512, 1024], True).to(device).eval()
    fp3 = PointNetFeaturePropagation(in_channel=1536, mlp=[256, 256]).to(device).eval()
    fp2 = PointNetFeaturePropagation(in_channel=576, mlp=[256, 128]).to(device).eval()
    with torch.no_grad():
        for m in list(sa3.mlp_bns) + list(fp3.mlp_bns) + list(fp2.mlp_bns):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    l1_xyz, l2_xyz = torch.rand(B, 3, 512, device=device) - 0.5, torch.rand(B, 3, 128, device=device) - 0.5
    l1_pts, l2_pts = torch.randn(B, 320, 512, device=device), torch.randn(B, 512, 128, device=device).relu()

    def run():
        with torch.no_grad(), fused.use_mlp_dtype("bf16"):
            l3_xyz, l3 = sa3(l2_xyz, l2_pts)
            l2n = fp3(l2_xyz, l3_xyz, l2_pts, l3)
            return l3, l2n, fp2(l1_xyz, l2_xyz, l1_pts, l2n)

    new = run()
    fused.USE_TILE_BF16 = False
    try:
        old = run()
    finally:
        fused.USE_TILE_BF16 = True
    for a, o, name in zip(new, old, ("sa3", "fp3", "fp2")):
        assert a.shape == o.shape and a.dtype == o.dtype == torch.float32, name
        scale = float(o.abs().max())
        d = (a - o).abs()
        assert float(d.max()) <= 3e-2 * scale and float(d.mean()) <= 5e-4 * scale, (name, float(d.max()) / scale, float(d.mean()) / scale)


@pytest.mark.parametrize("B,l", [(10, 4096), (3, 12000), (40, 1000)])
def test_head12_persistent_form_equals_one_tile_per_workgroup(device, B, l):
    """captra_head12_bf16 with more tiles than CUs runs persistent (a workgroup walks a contiguous run of tiles, biases and GroupNorm
    coefficients in LDS, the next tile's rows prefetched under the epilogue): the same arithmetic in the same order, so y2 and its
    statistics equal the one-tile-per-workgroup launch bit for bit -- runs that cross cloud boundaries and ragged last tiles included."""
    import ctypes
    from captra_amd import _lib, fused
    rng = np.random.default_rng(B + l)
    lin1 = fused.pack(_dev((rng.standard_normal((128, 512)) / 11.3).astype(np.float32), device), _dev(rng.standard_normal(512).astype(np.float32), device))
    lin2 = fused.pack(_dev((rng.standard_normal((512, 512)) / 22.6).astype(np.float32), device), _dev(rng.standard_normal(512).astype(np.float32), device))
    x = torch.randn(B, l, 128, device=device).to(torch.bfloat16)
    ab1 = torch.randn(B, 512, 2, device=device)
    res = []
    for v in (0, 1):
        _lib.lib().captra_tile_bf16_set_persistent(ctypes.c_int(v))
        try:
            res.append(fused.head12_bf16(x, lin1, ab1, lin2))
        finally:
            _lib.lib().captra_tile_bf16_set_persistent(ctypes.c_int(1))
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16)) and torch.equal(res[0][1], res[1][1])
