"""The backbone's neck in the bf16 mode as three launches (captra_neck_chain_bf16, csrc/neck_bf16.hip): SA3 (group_all set abstraction,
reference pointnet_utils.py:302-343), FP3 (feature propagation from ONE source vector per cloud, l.265-298) and FP2 (3-NN
interpolation + skip concat + shared MLP, l.280-298) must equal the layer-by-layer route they replace -- captra_dense_bf16_tile_ex
chains, captra_gemv_bf16, captra_interp_concat -- bit for bit: same MFMA sequence per layer, same bf16 roundings of the hidden
activations, same interpolation expression, same summation order of the per-cloud product."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _layers(dims, rng, device):
    from captra_amd import fused
    return [fused.pack(_dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device),
                       _dev(0.1 * rng.standard_normal(dims[i + 1]).astype(np.float32), device)) for i in range(len(dims) - 1)]


def _both(fn):
    """fn() with the one-launch module and layer by layer."""
    from captra_amd import fused
    fused.set_mlp_dtype("bf16")
    try:
        fused.USE_NECK_CHAIN = True
        got = fn()
        fused.USE_NECK_CHAIN = False
        want = fn()
    finally:
        fused.USE_NECK_CHAIN = True
        fused.set_mlp_dtype("fp32")
    torch.cuda.synchronize()
    return got, want


@pytest.mark.parametrize("B,N", [(1, 128), (3, 128), (32, 128), (2, 64)])
def test_sa3_group_all_one_launch(device, B, N):
    from captra_amd import _lib
    from captra_amd.pointnet_utils import PointNetSetAbstraction
    rng = np.random.default_rng(B + N)
    mod = PointNetSetAbstraction(None, None, None, 515, [256, 512, 1024], group_all=True).to(device).eval()
    mod._folded = _layers((515, 256, 512, 1024), rng, device)
    xyz = _dev(rng.random((B, 3, N), dtype=np.float32) - 0.5, device)
    feat = _dev(np.abs(rng.standard_normal((B, 512, N))).astype(np.float32), device)
    _lib.prof_reset(); _lib.prof_enable(True)
    got, want = _both(lambda: mod(xyz, feat)[1])
    _lib.prof_enable(False)
    assert _lib.prof_read("neck_chain")[1] == 1, "the one-launch kernel did not run"
    assert got.shape == (B, 1024, 1) and torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("B,N", [(1, 128), (3, 128), (32, 128), (2, 72)])
def test_fp3_one_source_vector_one_launch(device, B, N):
    from captra_amd import _lib
    from captra_amd.pointnet_utils import PointNetFeaturePropagation
    rng = np.random.default_rng(10 * B + N)
    mod = PointNetFeaturePropagation(1536, [256, 256]).to(device).eval()
    mod._folded = _layers((1536, 256, 256), rng, device)
    xyz1 = _dev(rng.random((B, 3, N), dtype=np.float32), device)
    xyz2 = torch.zeros(B, 3, 1, device=device)
    p1 = _dev(np.abs(rng.standard_normal((B, 512, N))).astype(np.float32), device)
    p2 = _dev(np.abs(rng.standard_normal((B, 1024, 1))).astype(np.float32), device)
    _lib.prof_reset(); _lib.prof_enable(True)
    got, want = _both(lambda: mod(xyz1, xyz2, p1, p2))
    _lib.prof_enable(False)
    assert _lib.prof_read("neck_chain")[1] == 1, "the one-launch kernel did not run"
    assert got.shape == (B, 256, N) and torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("B,N,S", [(1, 512, 128), (3, 512, 128), (32, 512, 128), (2, 200, 50)])
def test_fp2_interpolation_concat_and_layers_one_launch(device, B, N, S):
    from captra_amd import _lib
    from captra_amd.pointnet_utils import PointNetFeaturePropagation
    rng = np.random.default_rng(100 * B + N)
    mod = PointNetFeaturePropagation(576, [256, 128]).to(device).eval()
    mod._folded = _layers((576, 256, 128), rng, device)
    x1 = rng.random((B, N, 3), dtype=np.float32) - 0.5
    xyz1_n3 = _dev(x1, device)
    xyz2_n3 = _dev(x1[:, :S], device)
    xyz1, xyz2 = xyz1_n3.transpose(1, 2).contiguous(), xyz2_n3.transpose(1, 2).contiguous()
    p1 = _dev(np.abs(rng.standard_normal((B, 320, N))).astype(np.float32), device)
    p2 = _dev(np.abs(rng.standard_normal((B, 256, S))).astype(np.float32), device)
    _lib.prof_reset(); _lib.prof_enable(True)
    got, want = _both(lambda: mod(xyz1, xyz2, p1, p2, xyz1_n3=xyz1_n3, xyz2_n3=xyz2_n3))
    _lib.prof_enable(False)
    assert _lib.prof_read("neck_chain")[1] == 1, "the one-launch kernel did not run"
    assert got.shape == (B, 128, N) and torch.equal(got, want), float((got - want).abs().max())
