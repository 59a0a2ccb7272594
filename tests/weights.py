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
