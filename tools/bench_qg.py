"""captra_query_and_group (the reference's QueryAndGroup module as one launch) on the workload's five (level, radius) shapes, both
networks' calls, against captra_ball_query + captra_group_points (+ torch glue); us per call, and a sweep of the kernel's shape knobs."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from captra_amd import _lib, fused  # noqa: E402
from captra_amd import synthetic as clouds  # noqa: E402
from captra_amd.pointnet_lib import pointnet2_utils as pn  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sweep = "--sweep" in sys.argv
pts = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)])).to(dev).contiguous()


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


lib = _lib.lib()
tot_b = tot_t = 0.0
for (n, m, r, k, c) in [(4096, 512, 0.05, 32, 0), (4096, 512, 0.1, 64, 0), (4096, 512, 0.2, 128, 0), (4096, 512, 0.05, 32, 3), (4096, 512, 0.1, 64, 3),
                        (4096, 512, 0.2, 128, 3), (512, 128, 0.2, 64, 320), (512, 128, 0.4, 128, 320)]:
    xyz = pts[:, :n].contiguous()
    new = xyz[:, :m].contiguous()
    feat = torch.randn(B, c, n, device=dev) if c else None
    nb = B * (12.0 * n + 12.0 * m + 4.0 * m * k + sum(4.0 * cc * n + 4.0 * m * k + 4.0 * cc * m * k for cc in ((3, c) if c else (3,))))

    def two_ops():
        idx = pn.ball_query(r, k, xyz, new)
        gx = pn.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new.transpose(1, 2).unsqueeze(-1)
        return gx if feat is None else torch.cat([pn.grouping_operation(feat, idx), gx], dim=1)

    def one():
        return fused.query_and_group(r, k, xyz, new, feat, True)

    assert torch.equal(one(), two_ops())
    lib.captra_query_and_group_set_shape(0, 0)
    t1, t2 = timed(one), timed(two_ops)
    line = f"n={n} m={m} r={r} k={k} c={c}: one launch {t1:7.1f} us = {nb / t1 / 1e6:6.2f} TB/s ({nb / t1 / 8e6:.3f} of 8), two ops + glue {t2:7.1f} us"
    if sweep:
        if c < 8:
            combos = [(mcb, nt, 0, 0) for mcb in (16, 32) for nt in (256, 512)]
        else:
            combos = [(mcb, 256, cc, cs) for mcb in (16, 32) for cc in (8, 16) for cs in (cc, 48, 96, 192, 323)]
        for mcb, nt, cc, cs in combos:
            if mcb * k > 8192:
                continue
            lib.captra_query_and_group_set_shape(mcb | (nt << 8), cc | (cs << 8))
            try:
                line += f" | mcb {mcb} nt {nt} cc {cc} cs {cs}: {timed(one):6.1f}"
            except Exception as e:
                line += f" | mcb {mcb} nt {nt} cc {cc} cs {cs}: {type(e).__name__}"
        lib.captra_query_and_group_set_shape(0, 0)
    print(line, flush=True)
    tot_b += nb
    tot_t += t1
print(f"all eight calls: {tot_t:7.1f} us, {tot_b / tot_t / 1e6:.2f} TB/s = {tot_b / tot_t / 8e6:.3f} of the 8 TB/s spec (SA2 calls counted once: a frame has them twice)")
