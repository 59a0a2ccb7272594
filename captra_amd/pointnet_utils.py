"""Point-set functions and modules of the PointNet++ backbone, on HIP kernels.

Mirrors the public surface of the reference's network/models/pointnet_utils.py: the functions
`knn_point` (l.12), `three_nn` (l.35), `three_interpolate` (l.46), `gather_operation` (l.100),
`group_operation` (l.106), `farthest_point_sample` (l.112), `query_ball_point` (l.141) and the
modules `PointNetSetAbstractionMsg` (l.191), `PointNetFeaturePropagation` (l.253),
`PointNetSetAbstraction` (l.302) with identical constructor arguments and parameter names, so a
reference state dict loads unchanged.

Differences by design:
  * there is no import-time CUDA/CPU switch (reference l.8-10): every function runs the HIP
    operator and raises on CPU tensors;
  * `gather_operation` / `group_operation` call the kernels (the reference bypasses its own CUDA
    kernels with torch indexing, l.100-109);
  * `three_nn` follows the CUDA semantics (sqrt of the squared distance, pointnet2_utils.py:134),
    which is what the released checkpoints were trained with;
  * in eval mode the modules run a FUSED path: BatchNorm folded into the conv weights once,
    group-and-concat fused into the first layer's operand load, max-over-K fused into the last
    layer's epilogue, all three radii of a level served by one ball-query scan.  In train mode
    they run layer by layer with autograd (torch convs over the HIP grouping op).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .fold import fold_conv_bn
from .pointnet_lib import pointnet2_utils as futils


# ---------------------------------------------------------------------------------------------
# functions (same names / argument order as the reference)
# ---------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """(B,N,C), (B,M,C) -> (B,N,M) squared distances in the expanded form |a|^2 + |b|^2 - 2ab (reference l.56-78; plain
    torch, any device: the reference's CPU path builds its neighbour searches on it — the HIP operators use the direct form)."""
    dist = -2.0 * torch.matmul(src, dst.transpose(1, 2))
    dist = dist + torch.sum(src ** 2, dim=-1).unsqueeze(-1)
    return dist + torch.sum(dst ** 2, dim=-1).unsqueeze(1)


def knn_point(k, pos2, pos1):
    """k nearest of pos1 (B,N,3) for every query pos2 (B,M,3) -> (L2 distances (B,M,k), idx long)."""
    val, idx = futils.knn(k, pos2, pos1)
    return val, idx.long()


def three_nn(xyz1, xyz2):
    """3 nearest of xyz2 (B,M,3) for every xyz1 (B,N,3) -> (distances (B,N,3), idx long)."""
    dists, idx = futils.three_nn(xyz1, xyz2)
    return dists, idx.long()


def three_interpolate(points, idx, weight):
    """points (B,C,M), idx (B,N,3), weight (B,N,3) -> (B,C,N)."""
    return futils.three_interpolate(points, idx.int(), weight)


def gather_operation(feature, idx):
    """feature (B,C,N), idx (B,npoint) -> (B,C,npoint)."""
    return futils.gather_operation(feature, idx.int())


def group_operation(feature, idx):
    """feature (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
    return futils.grouping_operation(feature, idx.int())


def farthest_point_sample(xyz, npoint):
    """xyz (B,N,3) -> (B,npoint) long; starts at index 0 like the CUDA kernel (sampling_gpu.cu:113)."""
    return futils.furthest_point_sample(xyz, npoint).long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,S,3) -> (B,S,nsample) long."""
    return futils.ball_query(radius, nsample, xyz, new_xyz).long()


def index_points(points, idx):
    """points (B,N,C), idx (B,S...) -> (B,S...,C) (pure indexing helper)."""
    B = points.shape[0]
    view = [B] + [1] * (idx.dim() - 1)
    batch = torch.arange(B, dtype=torch.long, device=points.device).view(view).expand_as(idx)
    return points[batch, idx, :]


def sample_and_group_all(xyz, points):
    """xyz (B,N,3), points (B,N,D) -> (new_xyz zeros (B,1,3), (B,1,N,3+D)) — xyz first."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device, dtype=xyz.dtype)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


# ---------------------------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------------------------
class _FoldCache:
    """Mixin: folded (conv+BN) weights, rebuilt lazily after load_state_dict()/train()/to()."""

    def _invalidate(self):
        if getattr(self, "_folded", None) is not None:
            from .fold import bump_weights_version
            bump_weights_version()
        self._folded = None

    def train(self, mode: bool = True):
        # eval() on a model already in eval mode (every Trainer.test call) keeps the folded weights: BatchNorm statistics
        # only move in training mode, and a captured hipGraph may hold these tensors' addresses
        if mode != self.training:
            self._invalidate()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._invalidate()
        return super()._load_from_state_dict(*a, **k)


def _has_points(points):
    return points is not None and points.shape[1] > 0


MULTI_SCALE_MAX_CLOUDS = int(__import__("os").environ.get("CAPTRA_MULTI_SCALE", "2"))   # 0 = a level's scales always one launch each


class PointNetSetAbstractionMsg(_FoldCache, nn.Module):
    """Multi-scale-grouping set abstraction (reference pointnet_utils.py:191-250)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, knn=False):
        super().__init__()
        self.npoint = npoint
        self.radius_list = list(radius_list)
        self.nsample_list = list(nsample_list)
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        self.out_channel = 0
        for widths in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel
            for w in widths:
                convs.append(nn.Conv2d(last, w, 1))
                bns.append(nn.BatchNorm2d(w))
                last = w
            self.out_channel += last
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self.knn = knn
        self._folded = None

    def _fold(self, device):
        if self._folded is None:
            self._folded = [[fold_conv_bn(c, b, device) for c, b in zip(convs, bns)]
                            for convs, bns in zip(self.conv_blocks, self.bn_blocks)]
        return self._folded

    def window_ok(self, cfeat: int) -> bool:
        """Every scale of this level runs on a kernel that honours a centre window (the small-input scales of the CAPTRA
        backbone: fused.SA_WINDOW_SHAPES)."""
        folded = self._folded
        if folded is None or cfeat > 3:
            return False
        return all((cfeat, l[0].cout, l[1].cout, l[2].cout, k) in fused.SA_WINDOW_SHAPES for l, k in zip(folded, self.nsample_list))

    def _can_fuse(self, xyz):
        return (not self.training) and (not self.knn) and xyz.is_cuda and \
            all(k % 32 == 0 and 128 % k == 0 for k in self.nsample_list)

    def forward(self, xyz, points, xyz_n3=None, geom=None):
        """xyz (B,3,N), points (B,D,N) or None -> (new_xyz (B,3,S), features (B,D',S)).
        `xyz_n3` optionally passes the (B,N,3) copy the caller already has; `geom` the sampling and
        neighbour lists another network computed on the SAME cloud (FPS and ball query depend on the
        coordinates only).  The geometry used is left in `self.last_geom`."""
        if not _has_points(points):
            points = None
        if xyz_n3 is None:
            xyz_n3 = xyz.transpose(1, 2).contiguous()
        if geom is None:
            sampled = fused.fps_gather(xyz_n3.contiguous(), self.npoint) if self._can_fuse(xyz) else None
            if sampled is not None:                                                          # one launch
                _, new_xyz_n3, new_xyz_cn = sampled
                geom = {"new_xyz_n3": new_xyz_n3, "new_xyz": new_xyz_cn, "idx_list": None}
            else:
                fps_idx = futils.furthest_point_sample(xyz_n3, self.npoint)                  # (B,S) int32
                new_xyz_n3 = torch.gather(xyz_n3, 1, fps_idx.long().unsqueeze(-1).expand(-1, -1, 3))  # (B,S,3)
                geom = {"new_xyz_n3": new_xyz_n3, "new_xyz": new_xyz_n3.transpose(1, 2).contiguous(), "idx_list": None}
        new_xyz_n3, new_xyz = geom["new_xyz_n3"], geom["new_xyz"]
        self.last_new_xyz_n3 = new_xyz_n3
        self.last_geom = geom
        pooled = geom.get("pooled")
        if pooled is not None and id(self) in pooled:
            # the level-1 stream kernel (fused.sa1_stream_bf16) ran this level beside the sampler: nothing left to launch
            return new_xyz, pooled[id(self)]
        if self._can_fuse(xyz):
            return new_xyz, self._forward_fused(xyz.contiguous(), xyz_n3, points, new_xyz_n3, geom)
        return new_xyz, self._forward_layers(xyz, xyz_n3, points, new_xyz, new_xyz_n3)

    # eval: fused kernels
    def _forward_fused(self, xyz_cn, xyz_n3, points, new_xyz_n3, geom):
        folded = self._fold(xyz_cn.device)
        B, S = xyz_cn.shape[0], self.npoint
        if geom["idx_list"] is None:
            geom["idx_list"] = fused.ball_query_multi(self.radius_list, self.nsample_list, xyz_n3, new_xyz_n3)
        idx_list = geom["idx_list"]
        out = torch.empty(B, self.out_channel, S, dtype=torch.float32, device=xyz_cn.device)
        feat = points.contiguous() if points is not None else None
        if geom.get("chunks"):
            # STREAMED sampling (backbones.precompute_geometry_streamed): the sampler is still picking the later centres on its
            # own stream; every window of centres runs its scales as soon as its picks and neighbour lists are there
            cur = torch.cuda.current_stream(xyz_cn.device)
            cfeat = 0 if feat is None else feat.shape[1]
            for m0, mc, ev in geom["chunks"]:
                cur.wait_event(ev)
                off = 0
                with fused.centre_window(m0, mc):
                    for layers, idx in zip(folded, idx_list):
                        if fused.sa_scale_bf16_supported(cfeat, layers, idx.shape[2]):
                            fused.sa_scale_bf16(feat, xyz_cn, new_xyz_n3, idx, layers, out, off)
                        else:
                            fused.sa_scale_fused(feat, xyz_cn, new_xyz_n3, idx, layers, out, off)
                        off += layers[-1].cout
            return out
        off = 0
        prezero = B <= 8 and fused.exact_path() and len(folded) > 1
        if prezero:
            # few clouds: the fp32 scales run a wave per neighbour slice and combine a centre's slices by an atomic max on a zeroed
            # output -- ONE fill of the level's tensor here instead of a fill per scale and cloud in the launchers
            out.zero_()
        # ... and the level's scales handed over TOGETHER (captra_sa_scales_multi -> sa_wave_lds3_kernel / sa_wave_pipe2_kernel: each
        # scale on its own range of workgroups, bits unchanged) where a scale's own launch fills a fraction of the chip
        jobs = [] if (prezero and B <= MULTI_SCALE_MAX_CLOUDS) else None
        keep = []     # the scales' temporaries (v1): the job table holds raw pointers to them -- alive until the launch
        with fused.L.launch_options(sa_prezeroed=1 if prezero else 0):
            res = self._forward_scales(folded, idx_list, feat, xyz_cn, new_xyz_n3, out, off, B, keep, jobs)
            if jobs:
                fused.sa_scales_multi(jobs, xyz_cn.device)
        keep.clear()
        return res

    def _forward_scales(self, folded, idx_list, feat, xyz_cn, new_xyz_n3, out, off, B, keep=None, jobs=None):
        for layers, idx in zip(folded, idx_list):
            if fused.sa_scale_x6_supported(0 if feat is None else feat.shape[1], layers, idx.shape[2]):
                fused.sa_scale_x6(feat, xyz_cn, new_xyz_n3, idx, layers, out, off)      # cfg['mlp_dtype'] = "f32x6"
                off += layers[-1].cout
                continue
            if fused.sa_scale_bf16_supported(0 if feat is None else feat.shape[1], layers, idx.shape[2]):
                fused.sa_scale_bf16(feat, xyz_cn, new_xyz_n3, idx, layers, out, off)
                off += layers[-1].cout
                continue
            if feat is not None and fused.exact_path() and fused.sa_scale_pipe_supported(feat.shape[1], layers, idx.shape[1], idx.shape[2], b=B, n=feat.shape[2]):
                v1pm = fused.sa_first_layer_pre_pm(feat, layers[0])  # (B,N,c1) point-major: one 16-byte gather per register quad
                if keep is not None:
                    keep.append(v1pm)
                fused.sa_scale_pre_pm(v1pm, xyz_cn, new_xyz_n3, idx, layers, out, off, feat.shape[1], jobs=jobs)
                off += layers[-1].cout
                continue
            if feat is not None and fused.sa_scale_pre_supported(feat.shape[1], layers, idx.shape[2]):
                v1 = fused.sa_first_layer_pre(feat, layers[0])      # (B,c1,N): once per source point, not per neighbour
                if keep is not None:
                    keep.append(v1)
                fused.sa_scale_pre(v1, xyz_cn, new_xyz_n3, idx, layers, out, off, feat.shape[1])
                off += layers[-1].cout
                continue
            if fused.sa_scale_fusable(idx.shape[2], layers):
                fused.sa_scale_fused(feat, xyz_cn, new_xyz_n3, idx, layers, out, off, jobs=jobs)
                off += layers[-1].cout
                continue
            y = fused.sa_group_mlp(feat, xyz_cn, new_xyz_n3, idx, layers[0])
            for lin in layers[1:-1]:
                y = fused.pointwise_mlp(y, lin, fused.ACT_RELU)
            fused.mlp_max(y, layers[-1], out, off)
            off += layers[-1].cout
        return out

    # train / generic: layer by layer, differentiable
    def _forward_layers(self, xyz, xyz_n3, points, new_xyz, new_xyz_n3):
        B, C, N = xyz.shape
        S = self.npoint
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            if self.knn:
                _, group_idx = futils.knn(K, new_xyz_n3, xyz_n3)
            else:
                group_idx = futils.ball_query(radius, K, xyz_n3, new_xyz_n3)
            grouped_xyz = futils.grouping_operation(xyz, group_idx) - new_xyz.view(B, C, S, 1)
            if points is not None:
                grouped = torch.cat([futils.grouping_operation(points, group_idx), grouped_xyz], dim=1)
            else:
                grouped = grouped_xyz
            for conv, bn in zip(self.conv_blocks[i], self.bn_blocks[i]):
                grouped = F.relu(bn(conv(grouped)))
            outs.append(grouped.max(dim=-1)[0])
        return torch.cat(outs, dim=1)


class PointNetFeaturePropagation(_FoldCache, nn.Module):
    """3-NN inverse-distance interpolation + skip concat + shared MLP (reference l.253-299)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for w in mlp:
            self.mlp_convs.append(nn.Conv1d(last, w, 1))
            self.mlp_bns.append(nn.BatchNorm1d(w))
            last = w
        self.out_channel = last
        self._folded = None

    def _fold(self, device):
        if self._folded is None:
            self._folded = [fold_conv_bn(c, b, device) for c, b in zip(self.mlp_convs, self.mlp_bns)]
        return self._folded

    def forward(self, xyz1, xyz2, points1, points2, xyz1_n3=None, xyz2_n3=None, nn=None, tail=None, finish=None):
        """xyz1 (B,3,N) dense, xyz2 (B,3,S) sparse, points1 (B,D1,N) or None, points2 (B,D2,S)
        -> (B,D',N).  `nn` = (idx, weight) of fused.three_nn_weights on the same coordinates, when
        another network already computed them; what was used is left in `self.last_nn`.
        `tail` (fused path only) = a folded conv+BN+ReLU layer the caller applies to the result anyway
        (the backbone's conv1): it is appended to this module's MLP so that the chain runs as one launch.
        `finish(new_points, layers)` (fused path only), when given, replaces the evaluation of those layers: the caller
        runs them together with whatever consumes their output (CoordNet's heads) and gets back what it returns."""
        self.last_nn = None
        B, _, N = xyz1.shape
        S = xyz2.shape[2]
        if not _has_points(points1):
            points1 = None
        fuse = (not self.training) and xyz1.is_cuda
        if (S == 1 and fuse and fused.mlp_dtype() == "bf16" and points1 is not None and points2.shape[2] == 1 and finish is None
                and tail is None and self._fold(xyz1.device)[0].cout % 32 == 0):
            # bf16 mode, one source vector per cloud: W [x; v 1^T] + b = W1 x + (W2 v + b) -- the bracket is one small
            # product per cloud and becomes the first layer's bias; the (B, C1 + C2, N) repeat + concat is never built and the
            # layer runs on C1 instead of C1 + C2 input channels (1536 -> 512 for FP3)
            layers = list(self._fold(xyz1.device))
            c1 = points1.shape[1]
            lin0 = layers[0]
            if fused.neck_chain_supported(c1, N, [lin0.leading_rows(c1)] + layers[1:], c1, 1):
                # the whole module in one launch (csrc/neck_bf16.hip): the per-cloud product first, then both layers on LDS-resident tiles
                return fused.neck_chain(1, points1.contiguous(), [lin0.leading_rows(c1)] + layers[1:], v=points2.contiguous(),
                                        v_rows=lin0.trailing_rows(c1)).view(B, layers[-1].cout, N)
            if fused.chain_tile_bf16_supported(c1, N, layers):
                # LDS-tiled kernels (csrc/tile_bf16.hip): the per-cloud product as one small launch, then the layers
                bias_bc = fused.gemv_bf16(points2.contiguous(), lin0.trailing_rows(c1))
                sub = [lin0.leading_rows(c1)] + layers[1:]
                return fused.mlp_chain_bf16_tile(points1.contiguous(), sub, [fused.ACT_RELU] * len(sub), bias_bc=bias_bc).view(B, layers[-1].cout, N)
            bias_bc = fused.pointwise_mlp_bf16pm(points2.contiguous(), lin0.trailing_rows(c1), 1, in_pm=False, out_pm=False)
            y = fused.pointwise_mlp_bf16pm_cloud_bias(points1.contiguous(), lin0.leading_rows(c1), N, bias_bc.view(B, lin0.cout),
                                                      out_pm=len(layers) > 1, act=fused.ACT_RELU)
            for i, lin in enumerate(layers[1:]):
                y = fused.pointwise_mlp_bf16pm(y, lin, N, in_pm=True, out_pm=i < len(layers) - 2, act=fused.ACT_RELU)
            return y.view(B, layers[-1].cout, N)
        if S == 1 and fuse and points1 is not None and points2.shape[2] == 1 and finish is None and fused.exact_path():
            # one source vector per cloud (pointnet_utils.py:265-270): the first layer reads [points1; repeat(points2)] from the two
            # tensors as they are (captra_pointwise_mlp2: rows in the concat's order, so the same bits) -- no repeat, no concat
            layers = list(self._fold(xyz1.device)) + ([tail] if tail is not None else [])
            # f32x6 (no bit-exact contract): the repeated vector's share of the first layer once per cloud, as its bias
            y = fused.pointwise_mlp_cloud_bias(points1.contiguous(), points2.contiguous(), layers[0], fused.ACT_RELU)
            if y is None:
                y = fused.pointwise_mlp2(points1.contiguous(), points2.contiguous(), layers[0], fused.ACT_RELU)
            if y is not None:
                for lin in layers[1:]:
                    y = fused.pointwise_mlp(y, lin, fused.ACT_RELU)
                return y
        if S == 1:
            interpolated = points2.expand(-1, -1, N) if points2.shape[2] == 1 else points2.repeat(1, 1, N)
            new_points = torch.cat([points1, interpolated], dim=1) if points1 is not None else interpolated.contiguous()
        else:
            if xyz1_n3 is None:
                xyz1_n3 = xyz1.transpose(1, 2).contiguous()
            if xyz2_n3 is None:
                xyz2_n3 = xyz2.transpose(1, 2).contiguous()
            if fuse:
                if nn is None:
                    nn = fused.three_nn_weights(xyz1_n3, xyz2_n3)
                self.last_nn = nn
                if (points1 is not None and tail is None and finish is None and fused.mlp_dtype() == "bf16"
                        and fused.neck_chain_supported(points1.shape[1] + points2.shape[1], N, self._fold(xyz1.device), points1.shape[1], 2)):
                    # interpolation, skip concat and both layers in one launch (csrc/neck_bf16.hip): the interpolated channels are
                    # formed while the first layer's operand is staged
                    return fused.neck_chain(2, points1.contiguous(), self._fold(xyz1.device), x2=points2.contiguous(), nn=nn)
                new_points = fused.interp_concat(None if points1 is None else points1.contiguous(),
                                                 points2.contiguous(), nn[0], nn[1])
            else:
                dist, idx = futils.three_nn(xyz1_n3, xyz2_n3)
                recip = 1.0 / (dist + 1e-8)
                weight = recip / recip.sum(dim=2, keepdim=True)
                interpolated = futils.three_interpolate(points2, idx, weight)
                new_points = torch.cat([points1, interpolated], dim=1) if points1 is not None else interpolated
        if fuse:
            new_points = new_points.contiguous()
            layers = list(self._fold(xyz1.device)) + ([tail] if tail is not None else [])
            if finish is not None:
                return finish(new_points, layers)
            if len(layers) == 3:
                return fused.mlp_chain3(new_points, layers, fused.ACT_RELU)
            if fused.mlp_dtype() == "bf16":
                return fused.mlp_chain_bf16(new_points, layers, [fused.ACT_RELU] * len(layers))
            for lin in layers:
                new_points = fused.pointwise_mlp(new_points, lin, fused.ACT_RELU)
            return new_points
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = F.relu(bn(conv(new_points)))
        return new_points


class PointNetSetAbstraction(_FoldCache, nn.Module):
    """group_all set abstraction: concat [xyz, feat], shared MLP, max over all points (l.302-343)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for w in mlp:
            self.mlp_convs.append(nn.Conv2d(last, w, 1))
            self.mlp_bns.append(nn.BatchNorm2d(w))
            last = w
        self.out_channel = last
        self.group_all = group_all
        self.knn = knn
        self._folded = None

    def _fold(self, device):
        if self._folded is None:
            self._folded = [fold_conv_bn(c, b, device) for c, b in zip(self.mlp_convs, self.mlp_bns)]
        return self._folded

    def forward(self, xyz, points):
        """xyz (B,3,N), points (B,D,N) -> (new_xyz zeros (B,3,1), features (B,D',1))."""
        assert self.group_all, "only group_all is implemented (as in the reference, l.330)"
        B, C, N = xyz.shape
        if (not self.training) and xyz.is_cuda:
            # (eval: the all-zero "centre" of the pooled level is a constant nobody writes -- one tensor per shape, not a fill per step)
            # kept per shape and never freed while the module lives: a hipGraph captured at another batch size holds the address
            key = (B, C, xyz.device, xyz.dtype)
            cache = self.__dict__.setdefault("_zero_xyz", {})
            if key not in cache:
                cache[key] = torch.zeros(B, C, 1, device=xyz.device, dtype=xyz.dtype)
            new_xyz = cache[key]
        else:
            new_xyz = torch.zeros(B, C, 1, device=xyz.device, dtype=xyz.dtype)
        if (not self.training) and xyz.is_cuda and N % 32 == 0 and (128 % N == 0 or N % 128 == 0):
            # the fused max handles groups of 32 / 64 / 128 positions: a larger cloud is pooled as N/128 groups
            # of 128 and the group maxima are maxed again (max is exact, so the split does not change a bit)
            folded = self._fold(xyz.device)
            groups, k = (1, N) if N <= 128 else (N // 128, 128)
            if fused.mlp_dtype() == "bf16" and _has_points(points) and fused.neck_chain_supported(C + points.shape[1], N, folded, C, 0):
                return new_xyz, fused.neck_chain(0, xyz.contiguous(), folded, x2=points.contiguous())
            if fused.mlp_dtype() == "bf16" and _has_points(points) and fused.chain_tile_bf16_supported(C + points.shape[1], N, folded, pool=True):
                # LDS-tiled kernels: [xyz, feat] read as two sources (no concat), the max over the points in the last layer's epilogue
                return new_xyz, fused.mlp_chain_bf16_tile(xyz.contiguous(), folded, [fused.ACT_RELU] * len(folded), x2=points.contiguous(), pool=True)
            if fused.mlp_dtype() == "bf16":      # hidden activations bf16 point-major; the (exact) max on the last layer's fp32 output
                x = torch.cat([xyz, points], dim=1) if _has_points(points) else xyz   # (B,3+D,N), xyz first
                y = fused.mlp_chain_bf16(x.contiguous(), folded, [fused.ACT_RELU] * len(folded))
                return new_xyz, fused.row_max(y.view(B, self.out_channel, N))
            y = None
            if _has_points(points) and len(folded) > 1:
                # [xyz; feat] read from the two tensors as they are (captra_pointwise_mlp2): no concat, the same bits
                y = fused.pointwise_mlp2(xyz.contiguous(), points.contiguous(), folded[0], fused.ACT_RELU)
            if y is not None:
                y = y.view(B, folded[0].cout, groups, k)
                rest = folded[1:-1]
            else:
                x = torch.cat([xyz, points], dim=1) if _has_points(points) else xyz   # (B,3+D,N), xyz first
                y = x.contiguous().view(B, x.shape[1], groups, k)
                rest = folded[:-1]
            for lin in rest:
                y = fused.pointwise_mlp(y, lin, fused.ACT_RELU)
            out = torch.empty(B, self.out_channel, groups, dtype=torch.float32, device=xyz.device)
            fused.mlp_max(y, folded[-1], out, 0)
            return new_xyz, (out if groups == 1 else out.max(dim=2, keepdim=True)[0])
        x = torch.cat([xyz, points], dim=1) if _has_points(points) else xyz           # (B,3+D,N), xyz first
        y = x.unsqueeze(-1)                                                        # (B,3+D,N,1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            y = F.relu(bn(conv(y)))
        return new_xyz, y.max(dim=2)[0]
