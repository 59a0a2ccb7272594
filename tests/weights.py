"""Seeded synthetic weights, drawn in parameter-NAME order from numpy's default_rng so that the
reference (at golden-generation time) and captra_amd (at test time) load identical tensors
without any checkpoint being committed (SURVEY.md §7 step 1, §8c adjustment 5)."""
from __future__ import annotations

import numpy as np
import torch


def make_state_dict(shapes: dict, seed: int = 7) -> dict:
    """shapes: {state-dict key: tuple shape}.  Kaiming-like conv weights (activations keep their
    scale through ~20 layers), non-trivial BatchNorm statistics and affine parameters."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        if leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "weight" and len(shape) == 1:      # BatchNorm / GroupNorm gamma
            v = rng.uniform(0.7, 1.3, shape)
        elif leaf == "weight":                          # conv weight (cout, cin, 1[,1])
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        elif leaf == "bias":
            v = rng.normal(0.0, 0.05, shape)
        else:
            raise KeyError(f"unexpected state-dict leaf {name}")
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out


def shapes_of(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def make_physical_state_dict(shapes: dict, seed: int, num_parts: int, sym: bool, kind: str = "nocs", nocs_gain: float = 4.6,
                             rot_head_gain: float = 0.05, nocs_mix: float = 0.1) -> dict:
    """Seeded weights under which the track loop TRACKS the synthetic trajectories (positive, converging scales; a
    translation estimate that does not inherit the previous frame's error; small frame-to-frame rotations), so that a
    free-running trajectory does not amplify rounding noise and can be held to the 1e-4 contract on every frame (golden
    G9p), and the timed trajectories of bench.py stay meaningful over hundreds of frames.

    `make_state_dict(shapes, seed)` with plants (every other tensor keeps its random values and still feeds every output):
      * CoordinateNet carries nine numbers per point through identity rows (BatchNorm identity on those rows; all of them
        are >= 0 on |x| < 1, so the ReLUs pass them unchanged): (x + 1) of the canonicalised input coordinates, and the
        cloud's bounding box as max(x) + 1 and 1 - min(x) -- taken by the max-pools of SA1 (ball), SA2 (ball) and SA3
        (group_all) over the coordinate FEATURES this backbone groups (use_xyz_feat), broadcast back by FP3 and carried
        down by FP2 / FP1's interpolation (weights sum to 1).  The NOCS output conv combines them into the box-centred
        coordinate u = x - (max + min) / 2 with gain `nocs_gain`: sigmoid(g u) - 0.5 ~ u (slope g/4 > 1 at 0, < 1 far out,
        so the fitted scale has an attracting fixed point), plus `nocs_mix` x its random weights on all 128 channels.  A
        box-centred prediction does not depend on where the previous pose put the cloud: translation errors do not persist;
      * the segmentation head reads the box-centred y the same way on top of `nocs_mix` x its random weights: `kind` "nocs"
        (S-nocs clouds: object above its background) splits at u_y = -0.24, "arti" (S-arti: four boxes stacked along y)
        labels by the nearest box centre -- every part keeps hundreds of points and few points sit near a decision
        boundary (the generator asserts a margin);
      * RotationNet: the rotation heads' output conv is scaled by `rot_head_gain` and biased to the identity rotation
        ((0,1,0) for symmetric objects, ortho6d (1,0,0,0,1,0) otherwise): dR = I + a small input-dependent rotation.
    """
    sd = make_state_dict(shapes, seed)

    def plant(conv, bn, rows):
        """rows: [(out channel, {in channel: weight}, bias)] -- the row is zeroed first, its BatchNorm made the identity."""
        w = sd[f"{conv}.weight"]
        for o, ins, bias in rows:
            w[o] = 0.0
            for i, v in ins.items():
                w[o, i] = v
            sd[f"{conv}.bias"][o] = bias
            if bn is not None:
                sd[f"{bn}.weight"][o] = 1.0
                sd[f"{bn}.bias"][o] = 0.0
                sd[f"{bn}.running_mean"][o] = 0.0
                sd[f"{bn}.running_var"][o] = 1.0

    def carry(conv, bn, n, src0=0):
        plant(conv, bn, [(k, {src0 + k: 1.0}, 0.0) for k in range(n)])

    B = "npcs_net.backbone"
    # SA1, scale 0 (input: coordinate features 0..2, relative xyz 3..5): x + 1 and 1 - x, then max over the ball
    plant(f"{B}.sa1.conv_blocks.0.0", f"{B}.sa1.bn_blocks.0.0",
          [(c, {c: 1.0}, 1.0) for c in range(3)] + [(3 + c, {c: -1.0}, 1.0) for c in range(3)])
    carry(f"{B}.sa1.conv_blocks.0.1", f"{B}.sa1.bn_blocks.0.1", 6)
    carry(f"{B}.sa1.conv_blocks.0.2", f"{B}.sa1.bn_blocks.0.2", 6)
    # SA2, scale 0 (input: the 320 SA1 features first, relative xyz last): carry, max over the ball
    for l in range(3):
        carry(f"{B}.sa2.conv_blocks.0.{l}", f"{B}.sa2.bn_blocks.0.{l}", 6)
    # SA3 group_all (input: xyz FIRST, then the 512 SA2 features): carry, max over all 128 points = the box
    carry(f"{B}.sa3.mlp_convs.0", f"{B}.sa3.mlp_bns.0", 6, src0=3)
    carry(f"{B}.sa3.mlp_convs.1", f"{B}.sa3.mlp_bns.1", 6)
    carry(f"{B}.sa3.mlp_convs.2", f"{B}.sa3.mlp_bns.2", 6)
    # FP3 (input: 512 SA2 features, then the 1024 broadcast SA3 features), FP2 (320 SA1 features, then FP3's 256)
    carry(f"{B}.fp3.mlp_convs.0", f"{B}.fp3.mlp_bns.0", 6, src0=512)
    carry(f"{B}.fp3.mlp_convs.1", f"{B}.fp3.mlp_bns.1", 6)
    carry(f"{B}.fp2.mlp_convs.0", f"{B}.fp2.mlp_bns.0", 6, src0=320)
    carry(f"{B}.fp2.mlp_convs.1", f"{B}.fp2.mlp_bns.1", 6)
    # FP1 (input: xyz, xyz, then FP2's 128): channels 0..2 = x + 1 of the point itself, 3..8 = the box
    plant(f"{B}.fp1.mlp_convs.0", f"{B}.fp1.mlp_bns.0",
          [(c, {c: 1.0}, 1.0) for c in range(3)] + [(3 + k, {6 + k: 1.0}, 0.0) for k in range(6)])
    carry(f"{B}.fp1.mlp_convs.1", f"{B}.fp1.mlp_bns.1", 9)
    carry(f"{B}.conv1", f"{B}.bn1", 9)
    carry("npcs_net.nocs_head.0", "npcs_net.nocs_head.1", 9)

    def centred(j, gain):
        """weights on channels (x_j + 1, max_j + 1, 1 - min_j) and the bias of gain * (x_j - (max_j + min_j) / 2)."""
        return {j: gain, 3 + j: -0.5 * gain, 6 + j: 0.5 * gain}, -gain

    w, b = sd["npcs_net.nocs_head.3.weight"], sd["npcs_net.nocs_head.3.bias"]
    w *= nocs_mix
    b *= nocs_mix
    for p in range(num_parts):
        for j in range(3):
            ins, bias = centred(j, nocs_gain)
            w[3 * p + j, :9] = 0.0
            for i, v in ins.items():
                w[3 * p + j, i] = v
            b[3 * p + j] = bias
    w, b = sd["npcs_net.seg_head.0.weight"], sd["npcs_net.seg_head.0.bias"]
    w *= nocs_mix
    b *= nocs_mix
    if kind == "nocs":                     # logit(part 0) - logit(background) = K (u_y - y0)
        K, y0 = 40.0, -0.24
        ins, bias = centred(1, K)
        for i, v in ins.items():
            w[0, i] += v
        b[0] += bias - K * y0
    else:                                  # nearest box centre along y: logit_p = K (c_p u_y - c_p^2 / 2)
        K = 200.0
        centres = [(-0.3 + 0.2 * p) / 1.2 for p in range(num_parts)]
        mid = sum(centres) / len(centres)
        for p, c in enumerate(centres):
            c -= mid
            ins, bias = centred(1, K * c)
            for i, v in ins.items():
                w[p, i] += v
            b[p] += bias - K * c * c / 2
    ident = (0.0, 1.0, 0.0) if sym else (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
    for p in range(num_parts):
        key = f"net.regress_net.pose_pred.rtvec_head.{p}.model.9"
        sd[f"{key}.weight"] *= rot_head_gain
        sd[f"{key}.bias"] = torch.tensor(ident, dtype=torch.float32)
    return sd
