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
    """Seeded weights under which the track loop stays in its PHYSICAL regime on the synthetic trajectories (positive,
    slowly varying scales; small frame-to-frame rotations), so that a free-running trajectory does not amplify rounding
    noise and can be held to the 1e-4 contract on every frame (golden G9p).

    `make_state_dict(shapes, seed)` with three plants (every other tensor, i.e. the whole SA / FP stack, keeps its random
    values and still feeds every output):
      * CoordinateNet: channels 0..2 of FP1's two layers, conv1 and the NOCS head's hidden layer pass (x + 1) of the
        canonicalised input coordinates through (identity rows, BatchNorm identity; relu(x + 1) = x + 1 on |x| < 1);
        the NOCS output conv reads them with gain `nocs_gain` (sigmoid(g x) - 0.5 ~ x: slope g/4 > 1 at 0, < 1 far out, so
        the fitted scale has an attracting fixed point) plus `nocs_mix` x its random weights on all 128 channels;
      * the segmentation head reads the passed-through canonical y on top of `nocs_mix` x its random weights: `kind`
        "nocs" (S-nocs clouds: object vs background below it) splits at y = -0.30, "arti" (S-arti: four boxes stacked
        along y) labels by the nearest box centre -- every part keeps hundreds of points and few points sit near a
        decision boundary (the generator asserts a margin);
      * RotationNet: the rotation heads' output conv is scaled by `rot_head_gain` and biased to the identity rotation
        ((0,1,0) for symmetric objects, ortho6d (1,0,0,0,1,0) otherwise): dR = I + a small input-dependent rotation.
    """
    sd = make_state_dict(shapes, seed)

    def passthrough(conv, bn):
        w = sd[f"{conv}.weight"]
        w[:3] = 0.0
        for j in range(3):
            w[j, j] = 1.0
        sd[f"{conv}.bias"][:3] = 0.0
        sd[f"{bn}.weight"][:3] = 1.0
        sd[f"{bn}.bias"][:3] = 0.0
        sd[f"{bn}.running_mean"][:3] = 0.0
        sd[f"{bn}.running_var"][:3] = 1.0

    B = "npcs_net.backbone"
    passthrough(f"{B}.fp1.mlp_convs.0", f"{B}.fp1.mlp_bns.0")
    sd[f"{B}.fp1.mlp_convs.0.bias"][:3] = 1.0                 # (x + 1) >= 0 travels through the ReLUs unchanged
    passthrough(f"{B}.fp1.mlp_convs.1", f"{B}.fp1.mlp_bns.1")
    passthrough(f"{B}.conv1", f"{B}.bn1")
    passthrough("npcs_net.nocs_head.0", "npcs_net.nocs_head.1")
    w, b = sd["npcs_net.nocs_head.3.weight"], sd["npcs_net.nocs_head.3.bias"]
    w *= nocs_mix
    b *= nocs_mix
    for p in range(num_parts):
        for j in range(3):
            w[3 * p + j, :3] = 0.0
            w[3 * p + j, j] = nocs_gain
            b[3 * p + j] = -1.0 * nocs_gain
    w, b = sd["npcs_net.seg_head.0.weight"], sd["npcs_net.seg_head.0.bias"]
    w *= nocs_mix
    b *= nocs_mix
    if kind == "nocs":                     # logit(part 0) - logit(background) = K (y - y0), y + 1 travels in channel 1
        K, y0 = 40.0, -0.30
        w[0, 1] += K
        b[0] += -K * (y0 + 1.0)
    else:                                  # nearest box centre along y: logit_p = K (c_p y - c_p^2 / 2)
        K = 200.0
        centres = [(-0.3 + 0.2 * p) / 1.2 for p in range(num_parts)]
        mid = sum(centres) / len(centres)
        for p, c in enumerate(centres):
            c -= mid
            w[p, 1] += K * c
            b[p] += -K * c * 1.0 - K * c * c / 2
    ident = (0.0, 1.0, 0.0) if sym else (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
    for p in range(num_parts):
        key = f"net.regress_net.pose_pred.rtvec_head.{p}.model.9"
        sd[f"{key}.weight"] *= rot_head_gain
        sd[f"{key}.bias"] = torch.tensor(ident, dtype=torch.float32)
    return sd
