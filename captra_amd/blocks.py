"""1x1-conv MLP builders and the per-part rotation regressor.

Mirrors the used subset of the reference's network/models/blocks.py: `get_point_mlp`
(l.118-135), `MLPConv1d` (l.148-165), `RotationRegressor` (l.168-193).  Layers sit at the same
nn.Sequential indices as in the reference so that state-dict keys coincide
(`seg_head.0.weight`, `nocs_head.3.bias`, `rtvec_head.0.model.4.weight`, ...).
In eval mode on the GPU the 1x1 convs run through the fp32 MFMA kernel (BatchNorm folded).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .fold import fold_conv_bn
from .pose_utils.rotations import compute_rotation_matrix_from_ortho6d, normalize_vector

_ACTI = {"relu": lambda: nn.ReLU(inplace=True), "lrelu": lambda: nn.LeakyReLU(0.2, inplace=True),
         "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "softplus": nn.Softplus}


def _acti_layers(acti):
    return [] if acti == "none" else [_ACTI[acti]()]


def _norm_layers(norm, dim, channel_per_group=2):
    if norm == "bn":
        return [nn.BatchNorm1d(dim)]
    if norm == "gn":
        return [nn.GroupNorm(dim // channel_per_group, dim)]
    if norm == "in":
        return [nn.InstanceNorm1d(dim, affine=True)]
    assert norm == "none", norm
    return []


def get_conv_block(kernel_size, in_channels, out_channels, dropout=None, norm="none", acti="none"):
    """[conv, (dropout), (norm), (acti)] — 'valid' padding only (all the hot path uses)."""
    layers = [nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, bias=True)]
    if dropout is not None:
        layers.append(nn.Dropout(p=dropout))
    return layers + _norm_layers(norm, out_channels) + _acti_layers(acti)


def get_point_mlp(in_dim, out_dim, dims, acti="none", dropout=True, last_bn=False, keep_list=False):
    """1x1 convs; hidden layers BN + (dropout 0.5) + ReLU, last layer `acti`."""
    dropout = 0.5 if dropout else None
    dims = [in_dim] + list(dims) + [out_dim]
    layers = []
    for i in range(len(dims) - 2):
        layers += get_conv_block(1, dims[i], dims[i + 1], dropout=dropout, norm="bn", acti="relu")
    layers += get_conv_block(1, dims[-2], dims[-1], norm="bn" if last_bn else "none", acti=acti)
    return layers if keep_list else nn.Sequential(*layers)


def run_point_mlp(seq: nn.Sequential, x: torch.Tensor, cache: dict) -> torch.Tensor:
    """Evaluate a get_point_mlp / MLPConv1d Sequential on (B,C,N) through the fused kernels:
    Conv(+BatchNorm)(+ReLU | Sigmoid) groups become one MFMA launch.  Conv -> GroupNorm -> ReLU chains (the rotation
    heads) run without a normalisation pass: the conv's epilogue emits the group statistics and the next conv applies
    relu(a*x + b) while loading its operand (fused.USE_GN_FUSED); otherwise GroupNorm(+ReLU) is one separate kernel."""
    mods = list(seq)
    i = 0
    x = x.contiguous()
    if fused.mlp_dtype() == "bf16":
        y = _run_gn_chain_bf16(mods, x, cache)
        if y is not None:
            return y
    pending = None      # GroupNorm coefficients (B,C,2) of the layer that produced x, not applied yet (fused GN chain)
    while i < len(mods):
        conv = mods[i]
        assert isinstance(conv, nn.Conv1d), type(conv)
        j = i + 1
        while j < len(mods) and isinstance(mods[j], nn.Dropout):
            j += 1                                   # identity in eval mode
        bn = mods[j] if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d) else None
        if bn is not None:
            j += 1
        gn = mods[j] if j < len(mods) and isinstance(mods[j], nn.GroupNorm) else None
        if gn is not None:
            j += 1
        act = fused.ACT_NONE
        sigmoid_tail = False
        if j < len(mods) and isinstance(mods[j], nn.ReLU):
            act, j = fused.ACT_RELU, j + 1
        elif j < len(mods) and isinstance(mods[j], nn.Sigmoid):
            sigmoid_tail, j = True, j + 1
        key = id(conv)
        if key not in cache:
            cache[key] = fold_conv_bn(conv, bn, x.device)
        lin = cache[key]
        n_pos = x.numel() // (x.shape[0] * x.shape[1])
        if gn is not None and act == fused.ACT_RELU and fused.gn_chain_supported(x, lin.cout) and j < len(mods):
            # Conv -> GroupNorm -> ReLU with a consumer behind it: the conv emits the statistics, the consumer normalises
            x, stats = fused.pointwise_mlp_gn(x, lin, pending, fused.ACT_NONE, want_stats=True)
            pending = fused.gn_finalize(stats, gn.num_groups, gn.weight, gn.bias, gn.eps, n_pos)
            i = j
            continue
        if pending is not None:
            if gn is None and not sigmoid_tail:
                x = fused.pointwise_mlp_gn(x, lin, pending, act)
                pending = None
                i = j
                continue
            # a consumer the fused kernels do not cover: materialise the pending normalisation first
            x = torch.relu(x * pending[:, :, 0:1] + pending[:, :, 1:2])
            pending = None
        if gn is not None:
            x = fused.pointwise_mlp(x, lin, fused.ACT_NONE)
            cpg = x.shape[1] // gn.num_groups
            if x.shape[2] % 4 == 0 and cpg * x.shape[2] <= 16384:
                x = fused.group_norm_relu(x, gn.num_groups, gn.weight, gn.bias, gn.eps, relu=(act == fused.ACT_RELU))
            else:
                x = F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
                if act == fused.ACT_RELU:
                    x = F.relu(x, inplace=True)
        else:
            x = fused.pointwise_mlp(x, lin, act)
            if sigmoid_tail:
                x = torch.sigmoid(x)
        i = j
    if pending is not None:          # (cannot happen for the heads of this network: their last conv has no norm)
        x = torch.relu(x * pending[:, :, 0:1] + pending[:, :, 1:2])
    return x


def _run_gn_chain_bf16(mods, x, cache):
    """bf16 mode (cfg['mlp_dtype'] = 'bf16', BASELINE.json configs[2]): a Sequential of the exact form (Conv1d, GroupNorm,
    ReLU) x n, Conv1d -- the rotation heads (reference blocks.py:147-165) -- with the hidden activations kept in HBM as bf16
    point-major tensors (csrc/dense_bf16.hip): each hidden layer stores its raw output and, from its epilogue, the group statistics
    of what it stored, and the next layer applies relu(a x + b) while it loads its operand.  None when the Sequential has
    another shape (the caller's generic path runs it)."""
    layers, i = [], 0
    while i < len(mods):
        if not isinstance(mods[i], nn.Conv1d):
            return None
        if i + 2 < len(mods) and isinstance(mods[i + 1], nn.GroupNorm) and isinstance(mods[i + 2], nn.ReLU):
            layers.append((mods[i], mods[i + 1]))
            i += 3
        elif i == len(mods) - 1:
            layers.append((mods[i], None))
            i += 1
        else:
            return None
    pm_in = isinstance(x, fused.PMTensor)
    if pm_in:
        x, n, in_pm = x.data, x.data.shape[1], True
    else:
        n, in_pm = x.shape[2], False
    if len(layers) < 2 or not fused.gn_chain_bf16_supported(x, [c.out_channels for c, g in layers if g is not None]):
        return None
    ab = None
    for conv, gn in layers:
        if id(conv) not in cache:
            cache[id(conv)] = fold_conv_bn(conv, None, x.device)
    skip = 0
    if in_pm and len(layers) >= 3 and layers[0][1] is not None and layers[1][1] is not None \
            and fused.head12_bf16_supported(x, cache[id(layers[0][0])], cache[id(layers[1][0])]):
        # layers 1 + 2 in one launch (csrc/tile_bf16.hip): a statistics pass over y1 = W1 x + b1 (nothing stored), then y1 recomputed,
        # normalised in registers and consumed from LDS -- y1 (134 MB written + read at 32 x 4096 points) never reaches HBM
        (c1, g1), (c2, g2) = layers[0], layers[1]
        lin1, lin2 = cache[id(c1)], cache[id(c2)]
        ab1 = fused.gn_finalize(fused.head12_bf16_stats(x, lin1), g1.num_groups, g1.weight, g1.bias, g1.eps, n, tile_major=True)
        x, stats = fused.head12_bf16(x, lin1, ab1, lin2)
        ab = fused.gn_finalize(stats, g2.num_groups, g2.weight, g2.bias, g2.eps, n, tile_major=True)
        skip = 2
    for conv, gn in layers[skip:]:
        lin = cache[id(conv)]
        if in_pm and gn is not None and fused.dense_bf16_tile_supported(x, lin):
            x, stats = fused.dense_bf16_tile(x, lin, ab=ab, with_stats=True)      # LDS-tiled layer, statistics from its epilogue
            ab = fused.gn_finalize(stats, gn.num_groups, gn.weight, gn.bias, gn.eps, n, tile_major=True)
            continue
        epi = gn is not None and fused.USE_STATS_EPILOGUE   # the layer's epilogue leaves the statistics of what it stored
        if epi:
            x, stats = fused.pointwise_mlp_bf16pm(x, lin, n, in_pm=in_pm, out_pm=True, ab=ab, act=fused.ACT_NONE, with_stats=True)
        else:
            x = fused.pointwise_mlp_bf16pm(x, lin, n, in_pm=in_pm, out_pm=gn is not None, ab=ab, act=fused.ACT_NONE)
            stats = fused.gn_stats_bf16pm(x, lin.cout) if gn is not None else None
        if gn is not None:
            ab = fused.gn_finalize(stats, gn.num_groups, gn.weight, gn.bias, gn.eps, n, tile_major=epi)
            in_pm = True
    return x


class MLPConv1d(nn.Module):
    def __init__(self, in_channel, mlp, bn=True, gn=False, activation="relu", last_activation="none"):
        super().__init__()
        norm = "gn" if gn else ("bn" if bn else "none")
        layers, last = [], in_channel
        for i, width in enumerate(mlp):
            is_last = i == len(mlp) - 1
            layers += get_conv_block(1, last, width, norm="none" if is_last else norm,
                                     acti=last_activation if is_last else activation)
            last = width
        self.model = nn.Sequential(*layers)
        self.out_channel = last
        self._cache = {}

    def _drop_cache(self):
        if self._cache:
            from .fold import bump_weights_version
            bump_weights_version()
        self._cache = {}

    def train(self, mode=True):
        if mode != self.training:       # eval() while already in eval mode keeps the folded weights (see _FoldCache.train)
            self._drop_cache()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._drop_cache()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._drop_cache()
        return super()._apply(fn, *a, **k)

    def forward(self, input):
        if isinstance(input, fused.PMTensor):                # bf16 mode: the producer handed its output over point-major
            y = _run_gn_chain_bf16(list(self.model), input, self._cache)
            assert y is not None, "a point-major input needs the (Conv, GroupNorm, ReLU)*, Conv form"
            return y
        if (not self.training) and input.is_cuda:
            return run_point_mlp(self.model, input, self._cache)
        return self.model(input)


class RotationRegressor(nn.Module):
    """P heads, each 128 -> 512 -> 512 -> 256 -> (3 | 6) with GroupNorm(C/2); per-point output is
    normalised (symmetric: unit y-axis) or turned into a rotation matrix (ortho6d)."""

    def __init__(self, in_dim, num_parts, symmetric=False):
        super().__init__()
        rot_dim = 3 if symmetric else 6
        self.sym = symmetric
        self.rtvec_head = nn.ModuleList([MLPConv1d(in_dim, [512, 512, 256, rot_dim], bn=True, gn=True,
                                                   last_activation="none") for _ in range(num_parts)])
        self.num_parts = num_parts

    def raw(self, feat):
        """feat (B,in_dim,N) -> (B,P,R,N): the heads' raw outputs (R = 3 | 6), before the per-point normalisation."""
        if self.num_parts == 1:
            return self.rtvec_head[0](feat).unsqueeze(1)
        return torch.stack([head(feat) for head in self.rtvec_head], dim=1)

    def raw_diag(self, feat):
        """feat (B*P,in_dim,N), clouds ordered (trajectory, part) -> (B*P,R,N): head p evaluated on the clouds of part p
        only.  The tracking read-out keeps exactly these entries of the P x P evaluation (networks.py:200-203)."""
        P = self.num_parts
        if P == 1:
            return self.rtvec_head[0](feat)
        if isinstance(feat, fused.PMTensor):                 # (Q,N,C) point-major: the clouds of part p, contiguous, per head
            Q, N, C = feat.data.shape
            per_part = feat.data.view(Q // P, P, N, C)
            return torch.stack([self.rtvec_head[p](fused.PMTensor(per_part[:, p].contiguous(), feat.channels)) for p in range(P)],
                               dim=1).reshape(Q, -1, N)
        Q, C, N = feat.shape
        per_part = feat.view(Q // P, P, C, N)
        return torch.stack([self.rtvec_head[p](per_part[:, p].contiguous()) for p in range(P)], dim=1).reshape(Q, -1, N)

    def forward(self, feat):
        """feat (B,in_dim,N) -> (B,P,3,N) unit vectors or (B,P,9,N) row-major rotation matrices."""
        raw = self.raw(feat)                                                         # (B,P,R,N)
        per_point = raw.transpose(-1, -2)                                            # (B,P,N,R)
        shape = per_point.shape
        if self.sym:
            out = normalize_vector(per_point.reshape(-1, 3)).reshape(shape)
        else:
            out = compute_rotation_matrix_from_ortho6d(per_point.reshape(-1, 6)).reshape(shape[:-1] + (9,))
        return out.transpose(-1, -2)
