"""PointNet++ MSG encoder-decoder of CoordinateNet / RotationNet.

Mirrors `PointNet2Msg` of the reference's network/models/backbones.py:15-69: same constructor
(`cfg, out_dim, net_type, use_xyz_feat`), same sub-module names (sa1, sa2, sa3, fp3, fp2, fp1,
conv1, bn1) and therefore the same state-dict keys; forward [B,3(+C),N] -> [B,out_dim,N].
The point-major (B,N,3) copies that FPS / ball query / three_nn consume are produced once per
level and handed down instead of being re-derived by transposes inside every module.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .fold import fold_conv_bn
from .pointnet_utils import (PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg,
                             _FoldCache)


def _geom_tensors(geom):
    for v in geom.values():
        if torch.is_tensor(v):
            yield v
        elif isinstance(v, dict):
            yield from _geom_tensors(v)
        elif isinstance(v, (list, tuple)):
            for t in v:
                if torch.is_tensor(t):
                    yield t


class PointNet2Msg(_FoldCache, nn.Module):
    def __init__(self, cfg, out_dim, net_type="camera", use_xyz_feat=False):
        super().__init__()
        net_cfg = cfg["pointnet"][net_type]
        self.out_dim = out_dim
        self.in_dim = 3 if use_xyz_feat else 0
        self.use_xyz_feat = use_xyz_feat
        self.sa1 = PointNetSetAbstractionMsg(npoint=net_cfg["sa1"]["npoint"],
                                             radius_list=net_cfg["sa1"]["radius_list"],
                                             nsample_list=net_cfg["sa1"]["nsample_list"],
                                             in_channel=self.in_dim + 3,
                                             mlp_list=net_cfg["sa1"]["mlp_list"])
        self.sa2 = PointNetSetAbstractionMsg(npoint=net_cfg["sa2"]["npoint"],
                                             radius_list=net_cfg["sa2"]["radius_list"],
                                             nsample_list=net_cfg["sa2"]["nsample_list"],
                                             in_channel=self.sa1.out_channel + 3,
                                             mlp_list=net_cfg["sa2"]["mlp_list"])
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None,
                                          in_channel=self.sa2.out_channel + 3,
                                          mlp=net_cfg["sa3"]["mlp"], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=self.sa2.out_channel + self.sa3.out_channel,
                                              mlp=net_cfg["fp3"]["mlp"])
        self.fp2 = PointNetFeaturePropagation(in_channel=self.sa1.out_channel + self.fp3.out_channel,
                                              mlp=net_cfg["fp2"]["mlp"])
        self.fp1 = PointNetFeaturePropagation(in_channel=self.in_dim + 3 + self.fp2.out_channel,
                                              mlp=net_cfg["fp1"]["mlp"])
        self.conv1 = nn.Conv1d(self.fp1.out_channel, self.out_dim, 1)
        self.bn1 = nn.BatchNorm1d(self.out_dim)
        self.device = cfg["device"]
        self._folded = None

    def precompute_geometry(self, xyz_n3, level1_only=False, side=None, stream_level1=None):
        """Everything that depends on the coordinates only, in the layout `forward(geom=...)` takes: the two samplings, the
        ball-query lists of both levels, the 3-NN weights of FP1 / FP2.  Lets a caller run the MLP work of two networks
        that share a cloud on two streams (EvalTrackModel at small batch).  None when the fused samplers do not apply.
        `stream_level1` = (xyz_cn, planes or None, [(backbone, features (B,cf,N) or None), ...]): the backbones (one or two, this one
        among them) whose FIRST LEVEL runs inside the sampler's launch (fused.sa1_stream_bf16: ball query and shared MLPs of the
        centres picked so far on the CUs the sampler leaves idle); their pooled features ride in geom["sa1"]["pooled"]."""
        if self.training or not xyz_n3.is_cuda or self.sa1.knn or self.sa2.knn:
            return None
        pooled = lists = None
        if stream_level1 is not None:
            xyz_cn, planes, nets = stream_level1
            mods = [bb.sa1 for bb, _ in nets]
            for mod in mods:
                mod._fold(xyz_n3.device)
            cfs = [0 if f is None else f.shape[1] for _, f in nets]
            if fused.sa1_stream_supported(xyz_n3.shape[1], mods, cfs):
                m2 = self.sa2.npoint if (not level1_only and self.sa2.npoint <= 256) else 0
                res = fused.sa1_stream_bf16(xyz_n3.contiguous(), xyz_cn.contiguous(), mods, [f for _, f in nets], planes=planes, m2=m2)
                _, n1_n3, n1_cn, lists, outs, scratch = res[:6]
                pooled = {id(mod): out for mod, out in zip(mods, outs)}
                pooled["_scratch"] = scratch
                if m2:
                    pooled["_level2"] = res[6]
                first = (None, n1_n3, n1_cn)
        if pooled is None:
            first = fused.fps_gather(xyz_n3.contiguous(), self.sa1.npoint)
        if first is None:
            return None
        _, n1_n3, n1_cn = first
        if pooled is not None:
            g1 = {"new_xyz_n3": n1_n3, "new_xyz": n1_cn, "idx_list": lists, "pooled": pooled}
            if level1_only:
                return {"sa1": g1}
            if side is not None:
                main = torch.cuda.current_stream(xyz_n3.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    nn1 = fused.three_nn_weights(xyz_n3, n1_n3)
                geom = self.precompute_geometry_rest({"sa1": g1, "fp1": nn1}, xyz_n3)
                main.wait_stream(side)
                for t in nn1:
                    t.record_stream(main)
                return geom
            return self.precompute_geometry_rest({"sa1": g1}, xyz_n3)
        if side is not None and not level1_only:
            # everything below depends on the first sampling only: level 1's ball query and interpolation weights on `side`,
            # level 2's sampling, ball query and weights on this stream (a plain fork / join: ~0.06 ms off the serial prefix
            # both networks wait for)
            main = torch.cuda.current_stream(xyz_n3.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                idx1 = fused.ball_query_multi(self.sa1.radius_list, self.sa1.nsample_list, xyz_n3, n1_n3)
                nn1 = fused.three_nn_weights(xyz_n3, n1_n3)
            geom = {"sa1": {"new_xyz_n3": n1_n3, "new_xyz": n1_cn, "idx_list": idx1}, "fp1": nn1}
            geom = self.precompute_geometry_rest(geom, xyz_n3)
            main.wait_stream(side)
            for t in list(idx1) + list(nn1):
                t.record_stream(main)
            return geom
        g1 = {"new_xyz_n3": n1_n3, "new_xyz": n1_cn,
              "idx_list": fused.ball_query_multi(self.sa1.radius_list, self.sa1.nsample_list, xyz_n3, n1_n3)}
        geom = {"sa1": g1}
        if level1_only:
            return geom
        return self.precompute_geometry_rest(geom, xyz_n3)

    def precompute_geometry_streamed(self, xyz_n3, chunks: int, gstream, consumers=(), backbones=()):
        """`precompute_geometry` with the first level's sampling STREAMED: the 4096 -> 512 sampler is `chunks` launches of
        npoint / chunks picks each on `gstream` (captra_fps_gather_part: the same loop, cut), every part followed by the ball
        query of its centres and an event; `geom["sa1"]["chunks"]` = [(first centre, count, event)] is what `sa1` walks --
        a window's shared MLPs start when its picks and neighbour lists are there, while the sampler (one workgroup per cloud,
        511 dependent rounds: a quarter of a frame with nothing else able to run) picks the next ones.  Level 2 and the
        interpolation weights follow on `gstream`; `geom["_ready"]` is recorded behind them (forward waits for it before
        `sa2`).  Returns at once; None when a shape is outside the streamed kernels (caller: `precompute_geometry`).
        `gstream` must not be the current stream; `consumers`: the streams that will read the geometry (record_stream)."""
        if self.training or not xyz_n3.is_cuda or self.sa1.knn or self.sa2.knn or chunks < 2:
            return None
        M = self.sa1.npoint
        self.sa1._fold(xyz_n3.device)
        if M % chunks or (M // chunks) % 8 or not self.sa1.window_ok(self.in_dim):
            return None
        for bb in backbones:         # every backbone that will walk the windows (RotationNet's shares this geometry)
            bb.sa1._fold(xyz_n3.device)
            if not bb.sa1.window_ok(bb.in_dim) or bb.sa1.npoint != M:
                return None
        xyz_n3 = xyz_n3.contiguous()
        bufs = fused.fps_gather_parts(xyz_n3, M)
        if bufs is None:
            return None
        B = xyz_n3.shape[0]
        idx1 = [torch.empty(B, M, int(k), dtype=torch.int32, device=xyz_n3.device) for k in self.sa1.nsample_list]
        main = torch.cuda.current_stream(xyz_n3.device)
        gstream.wait_stream(main)
        step = M // chunks
        marks = []
        with torch.cuda.stream(gstream):
            for c in range(chunks):
                fused.fps_gather_part(xyz_n3, M, c * step, (c + 1) * step, bufs)
                with fused.centre_window(c * step, step):
                    fused.ball_query_multi(self.sa1.radius_list, self.sa1.nsample_list, xyz_n3, bufs[1], outs=idx1)
                ev = torch.cuda.Event()
                ev.record(gstream)
                marks.append((c * step, step, ev))
            geom = {"sa1": {"new_xyz_n3": bufs[1], "new_xyz": bufs[2], "idx_list": idx1, "chunks": marks}}
            geom = self.precompute_geometry_rest(geom, xyz_n3)
            ready = torch.cuda.Event()
            ready.record(gstream)
            geom["_ready"] = ready
        # the sampler's own buffers (picks, running minima: read and written by every later part) were allocated on the calling
        # stream and are used on `gstream` only: they stay alive with the geometry and are recorded on `gstream`, or the caching
        # allocator could hand them to the next allocation of the calling stream while the remaining parts are still queued
        geom["_keep"] = tuple(bufs) + tuple(idx1)
        for t in geom["_keep"]:
            t.record_stream(gstream)
        for t in _geom_tensors(geom):
            for st in (main,) + tuple(consumers):
                t.record_stream(st)
        return geom

    def precompute_geometry_rest(self, geom, xyz_n3):
        """Level 2 and the interpolation weights, given level 1 (`precompute_geometry(..., level1_only=True)`)."""
        n1_n3 = geom["sa1"]["new_xyz_n3"]
        lvl2 = (geom["sa1"].get("pooled") or {}).get("_level2")     # the level-1 stream kernel's samplers went on to level 2
        _, n2_n3, n2_cn = lvl2 if lvl2 is not None else fused.fps_gather(n1_n3, self.sa2.npoint)
        geom["sa2"] = {"new_xyz_n3": n2_n3, "new_xyz": n2_cn,
                       "idx_list": fused.ball_query_multi(self.sa2.radius_list, self.sa2.nsample_list, n1_n3, n2_n3)}
        if "fp1" not in geom:
            geom["fp1"] = fused.three_nn_weights(xyz_n3, n1_n3)
        geom["fp2"] = fused.three_nn_weights(n1_n3, n2_n3)
        return geom

    def forward(self, input, input_n3=None, geom=None, finish=None):
        """input (B,3(+C),N); `input_n3` optionally the (B,N,3) copy of input[:, :3]; `geom` optionally
        the `last_geom` of another PointNet2Msg that ran on the SAME cloud (FPS picks, ball-query lists
        and 3-NN weights depend on coordinates only): those kernels are then skipped."""
        geom = geom or {}
        l0_xyz = input[:, :3] if input.shape[1] > 3 else input
        l0_xyz = l0_xyz.contiguous()
        l0_points = input if self.use_xyz_feat else input[:, 3:]
        if input_n3 is None:
            input_n3 = l0_xyz.transpose(1, 2).contiguous()
        l1_xyz, l1_points = self.sa1(l0_xyz, l0_points, xyz_n3=input_n3, geom=geom.get("sa1"))
        l1_n3 = self.sa1.last_new_xyz_n3
        if geom.get("_ready") is not None:            # the rest of the geometry is being computed on another stream
            torch.cuda.current_stream(l0_xyz.device).wait_event(geom["_ready"])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, xyz_n3=l1_n3, geom=geom.get("sa2"))
        l2_n3 = self.sa2.last_new_xyz_n3
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)

        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points, xyz1_n3=l1_n3, xyz2_n3=l2_n3, nn=geom.get("fp2"))
        skip0 = torch.cat([l0_xyz, l0_points], dim=1) if l0_points.shape[1] > 0 else l0_xyz
        fuse = (not self.training) and l0_xyz.is_cuda
        if fuse and self._folded is None:
            self._folded = fold_conv_bn(self.conv1, self.bn1, l0_xyz.device)
        # fused path: conv1 + bn1 + ReLU rides at the end of FP1's MLP (one launch for the three layers)
        # `finish` (fused path): the caller evaluates FP1's layers + conv1 itself, fused with its own heads
        l0_points = self.fp1(l0_xyz, l1_xyz, skip0, l1_points, xyz1_n3=input_n3, xyz2_n3=l1_n3, nn=geom.get("fp1"),
                             tail=self._folded if fuse else None, finish=finish if fuse else None)
        self.last_geom = {"sa1": self.sa1.last_geom, "sa2": self.sa2.last_geom, "fp2": self.fp2.last_nn, "fp1": self.fp1.last_nn}
        if fuse:
            return l0_points
        return F.relu(self.bn1(self.conv1(l0_points)))
