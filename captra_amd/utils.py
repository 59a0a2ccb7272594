"""Small host helpers the track loop needs (mirrors the used subset of the reference's utils.py:
`Timer` l.139-152, `cvt_torch` l.175-185, `get_ith_from_batch` l.155-172, `ensure_dirs`)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch


class Timer:
    """Wall-clock ticks.  Unlike the reference (no sync before timing) `tick`
    synchronises the device first when `sync` is set, so GPU work is attributed correctly."""

    def __init__(self, on: bool = True, sync: bool = False):
        self.on, self.sync = on, sync
        self.cur = time.time()

    def tick(self, label=None):
        if not self.on:
            return None
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.time()
        diff, self.cur = now - self.cur, now
        if label is not None:
            print(label, diff)
        return diff


def ensure_dirs(paths):
    for p in paths if isinstance(paths, (list, tuple)) else [paths]:
        os.makedirs(p, exist_ok=True)


def cvt_torch(x, device):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float().to(device)
    if isinstance(x, torch.Tensor):
        return x.float().to(device)
    if isinstance(x, dict):
        return {k: cvt_torch(v, device) for k, v in x.items()}
    if isinstance(x, list):
        return [cvt_torch(v, device) for v in x]
    if x is None:
        return None
    raise TypeError(f"cvt_torch: unsupported type {type(x)}")


def get_ith_from_batch(data, i, to_single=True):
    if isinstance(data, dict):
        return {k: get_ith_from_batch(v, i, to_single) for k, v in data.items()}
    if isinstance(data, list):
        return [get_ith_from_batch(v, i, to_single) for v in data]
    if isinstance(data, torch.Tensor):
        return data[i].detach().cpu().item() if to_single else data[i].detach().cpu()
    if isinstance(data, np.ndarray):
        return data[i]
    if data is None or isinstance(data, str):
        return data
    raise TypeError(f"get_ith_from_batch: unsupported type {type(data)}")


def add_dict(total: dict, new: dict) -> None:
    """total += new, key by key, recursing into nested dicts (reference utils.py:46-70)."""
    for k, v in new.items():
        if isinstance(v, dict):
            add_dict(total.setdefault(k, {}), v)
        else:
            total[k] = total[k] + v if k in total else v


def divide_dict(d: dict, n: float) -> dict:
    return {k: divide_dict(v, n) if isinstance(v, dict) else v / n for k, v in d.items()}
