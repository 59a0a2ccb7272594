"""BackbonePipe (captra_amd/graph.py): PointNet2Msg.forward over a stream of independent batches with batch t + 1's geometry
(furthest-point sampling, ball queries, 3-NN weights) on one stream beside batch t's shared MLPs on another.  Every output
must be the plain forward's (reference network/models/backbones.py:30-69), bit for bit, whatever the interleaving."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("xyz_feat,depth", [(False, 2), (True, 3)])
def test_backbone_pipe_equals_plain_forward(device, xyz_feat, depth):
    from captra_amd import synthetic as clouds
    from captra_amd.backbones import PointNet2Msg
    from captra_amd.configs import make_config
    from captra_amd.graph import BackbonePipe
    from captra_amd.synthetic import make_state_dict
    cfg = copy.deepcopy(make_config("1"))
    net = PointNet2Msg(cfg, 128, use_xyz_feat=xyz_feat)
    net.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5))
    net = net.to(device).eval()
    B, N, T = 3, 4096, 7
    xs = [torch.from_numpy(np.ascontiguousarray(np.stack([clouds.s_nocs(100 * t + i)[0] for i in range(B)]).transpose(0, 2, 1))).to(device)
          for t in range(T)]
    with torch.no_grad():
        want = [net(x).clone() for x in xs]
    pipe = BackbonePipe(net, xs[0], depth=depth)
    got = []
    # outputs are read `depth - 1` pushes late, as a consumer of the pipeline would
    slots = []
    for t in range(T):
        slots.append(pipe.push(xs[t]))
        if t >= depth - 1:
            got.append(pipe.output(slots[t - (depth - 1)]).clone())
    for t in range(T - (depth - 1), T):
        got.append(pipe.output(slots[t]).clone())
    torch.cuda.synchronize()
    assert len(got) == T
    for t in range(T):
        assert torch.equal(got[t], want[t]), t
    # the resident input again
    s = pipe.push()                                   # slot T % depth still holds batch T - depth
    assert torch.equal(pipe.output(s), want[T - depth])
    pipe.drain()
    torch.cuda.synchronize()


def test_backbone_pipe_rejects_training_mode(device):
    from captra_amd.backbones import PointNet2Msg
    from captra_amd.configs import make_config
    from captra_amd.graph import BackbonePipe
    net = PointNet2Msg(copy.deepcopy(make_config("1")), 128).to(device).train()
    with pytest.raises(ValueError):
        BackbonePipe(net, torch.zeros(2, 3, 4096, device=device))
