"""Ball query at the SA1 / SA2 shapes: multi-radius scan, grid path on / off (captra_ball_query_set_prune).
Usage: python tools/bench_ball_query.py [--clouds 32]"""
import argparse
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import _lib  # noqa: E402
from captra_amd import synthetic as clouds  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.clouds
    pts = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)])).to(dev).contiguous()
    lib = _lib.lib()
    for (n, m, rk) in [(4096, 512, [(0.05, 32), (0.1, 64), (0.2, 128)]), (4096, 512, [(0.05, 32)]), (4096, 512, [(0.1, 64)]),
                       (4096, 512, [(0.2, 128)]), (512, 128, [(0.2, 64), (0.4, 128)])]:
        xyz = pts[:, :n].contiguous()
        new_xyz = xyz[:, :m].contiguous()
        nr = len(rk)
        outs = [torch.zeros(B, m, k, dtype=torch.int32, device=dev) for _, k in rk]
        radii = (ctypes.c_float * nr)(*[r for r, _ in rk])
        ks = (ctypes.c_int * nr)(*[k for _, k in rk])
        ptrs = (ctypes.c_void_p * nr)(*[o.data_ptr() for o in outs])

        def run():
            _lib.call("captra_ball_query_multi", B, n, m, nr, ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(ks, ctypes.c_void_p),
                      new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p))
        res = {}
        for prune in (0, 1):
            lib.captra_ball_query_set_prune(ctypes.c_int(prune))
            run()
            torch.cuda.synchronize()
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(a.iters):
                run()
            torch.cuda.synchronize()
            _lib.prof_enable(False)
            res[prune] = (sum(_lib.prof_read(nm)[0] for nm in _lib.prof_names()) / a.iters * 1e3, [o.clone() for o in outs])
        same = all(torch.equal(x, y) for x, y in zip(res[0][1], res[1][1]))
        print(f"N={n} M={m} radii {[r for r, _ in rk]}: scan {res[0][0]:7.1f} us   grid {res[1][0]:7.1f} us   identical lists: {same}", flush=True)
    lib.captra_ball_query_set_prune(ctypes.c_int(0))


if __name__ == "__main__":
    main()
