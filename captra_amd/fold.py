"""Fold eval-mode BatchNorm into the preceding 1x1 convolution (done once at load time).

y = gamma * (W x + b - mean) / sqrt(var + eps) + beta  ==  W' x + b'  with
W' = W * gamma / sqrt(var + eps),  b' = (b - mean) * gamma / sqrt(var + eps) + beta.
Computed in float64 on the host, stored as the fp32 TRANSPOSE W'^T (cin, cout) that the MFMA
kernels stage row-wise.  Replaces the separate Conv -> BatchNorm -> ReLU ATen calls of
pointnet_utils.py:242-245.
"""
from __future__ import annotations

import torch

# Bumped whenever a module drops its folded weights (mode change, load_state_dict, .to()).  Anything that baked raw device
# pointers of folded tensors into a captured hipGraph compares the value it was captured at (captra_amd/graph.py, model.py).
_WEIGHTS_VERSION = [0]


def weights_version() -> int:
    return _WEIGHTS_VERSION[0]


def bump_weights_version() -> None:
    _WEIGHTS_VERSION[0] += 1


def collect_folded(module) -> list:
    """Every PackedLinear currently cached under `module` (the `_folded` / `_cache` attributes of the fused modules):
    holding this list keeps the device buffers a captured graph points at alive."""
    found = []

    def walk(obj):
        if isinstance(obj, PackedLinear):
            found.append(obj)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                walk(o)
        elif isinstance(obj, dict):
            for o in obj.values():
                walk(o)

    for m in module.modules():
        walk(getattr(m, "_folded", None))
        walk(getattr(m, "_cache", None))
    return found


class PackedLinear:
    """A layer's weights in the layout the MFMA kernels take (include/captra_hip.h "PACKED WEIGHTS"): ONE buffer `wt` holding
    W'^T row-major, zero-padded to (ceil32(cin), ceil128(cout)), followed by the same numbers in MFMA-fragment order (the
    image the streaming kernels read with 16-byte loads); bias zero-padded to ceil128(cout).  `wt2d` is the row-major
    part as a (KP, CP) view."""

    __slots__ = ("wt", "wt2d", "bias", "cin", "cout", "_bf16")

    def __init__(self, wt_dense: torch.Tensor, bias_dense: torch.Tensor):
        cin, cout = wt_dense.shape
        kp, cp = (cin + 31) // 32 * 32, (cout + 127) // 128 * 128
        kq, nt = ((cin + 1) // 2 + 3) // 4, (cout + 31) // 32
        dev = wt_dense.device
        buf = torch.zeros(kp * cp + nt * kq * 256, dtype=torch.float32, device=dev)
        wt2d = buf[:kp * cp].view(kp, cp)
        wt2d[:cin, :cout] = wt_dense
        bias = torch.zeros(cp, dtype=torch.float32, device=dev)
        bias[:cout] = bias_dense
        self.wt, self.wt2d, self.bias, self.cin, self.cout = buf, wt2d, bias.contiguous(), cin, cout
        self._bf16 = {}
        if dev.type == "cuda":       # (a CPU-side PackedLinear only exists in host-logic tests: no kernel will read it)
            from . import _lib as L
            with torch.cuda.device(dev):
                L.call("captra_pack_weights_frag", cin, cout, L.ptr(buf))

    def leading_rows(self, rows: int) -> "PackedLinear":
        """The layer restricted to its first `rows` input channels (same bias): its own packed buffer (the fragment image
        depends on the row count), built once and cached with the layer."""
        assert 1 <= rows <= self.cin
        key = ("lead", rows)
        if key not in self._bf16:
            self._bf16[key] = PackedLinear(self.wt2d[:rows, :self.cout].contiguous(), self.bias[:self.cout].contiguous())
        return self._bf16[key]

    def trailing_rows(self, row0: int) -> "PackedLinear":
        """The layer restricted to its input channels from `row0` on (same bias), cached with the layer."""
        assert 0 <= row0 < self.cin
        key = ("trail", row0)
        if key not in self._bf16:
            self._bf16[key] = PackedLinear(self.wt2d[row0:self.cin, :self.cout].contiguous(), self.bias[:self.cout].contiguous())
        return self._bf16[key]

    def bf16(self, row0: int = 0, rows: int | None = None) -> torch.Tensor:
        """bf16 image of input rows [row0, row0+rows) of this layer for the bf16 kernels (include/captra_hip.h:
        Wb [ceil32(cout)][ceil32(rows)], untransposed, zero padded), built once per (row0, rows) on the device."""
        from . import _lib as L
        rows = self.cin - row0 if rows is None else rows
        key = (row0, rows)
        cache = self._bf16
        if key not in cache:
            dense = self.wt2d[row0:row0 + rows, :self.cout].contiguous()
            kb, cp = (rows + 31) // 32 * 32, (self.cout + 31) // 32 * 32
            wb = torch.empty(cp, kb, dtype=torch.bfloat16, device=self.wt.device)
            with torch.cuda.device(self.wt.device):
                L.call("captra_pack_weights_bf16", rows, self.cout, L.ptr(dense), L.ptr(wb))
            cache[key] = wb
        return cache[key]

    def bf16_frag(self, perm: bool) -> torch.Tensor:
        """Fragment-ordered bf16 image of the whole layer for the bf16-native dense kernel (captra_pack_dense_bf16); perm: the
        layer's input is a point-major slot-order tensor."""
        from . import _lib as L
        key = ("frag", bool(perm))
        if key not in self._bf16:
            img = torch.empty(L.lib().captra_dense_bf16_image_bytes(self.cin, self.cout), dtype=torch.uint8, device=self.wt.device)
            with torch.cuda.device(self.wt.device):
                L.call("captra_pack_dense_bf16", self.cin, self.cout, 1 if perm else 0, L.ptr(self.wt), L.ptr(img))
            self._bf16[key] = img
        return self._bf16[key]


def fold_conv_bn(conv, bn=None, device=None) -> PackedLinear:
    """conv: nn.Conv1d/Conv2d with 1x1 kernel; bn: BatchNorm or None -> PackedLinear on `device`."""
    w = conv.weight.detach().double().reshape(conv.weight.shape[0], -1).cpu()      # (cout, cin)
    b = conv.bias.detach().double().cpu() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    if bn is not None:
        g = bn.weight.detach().double().cpu() if bn.weight is not None else torch.ones_like(b)
        beta = bn.bias.detach().double().cpu() if bn.bias is not None else torch.zeros_like(b)
        inv = g / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        w = w * inv.unsqueeze(1)
        b = (b - bn.running_mean.detach().double().cpu()) * inv + beta
    dev = device if device is not None else conv.weight.device
    return PackedLinear(w.t().contiguous().float().to(dev), b.float().to(dev))
