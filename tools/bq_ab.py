"""Ball query, one vs two centres per wave (captra_ball_query_set_cpw), SA1 / SA2 multi-radius shapes at several batch sizes."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from captra_amd import _lib
from captra_amd import synthetic as clouds
dev = torch.device('cuda:0')
lib = _lib.lib()
for B in (1, 16, 32, 64):
    pts = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)])).to(dev).contiguous()
    for (n, m, rk) in [(4096, 512, [(0.05, 32), (0.1, 64), (0.2, 128)]), (512, 128, [(0.2, 64), (0.4, 128)])]:
        xyz = pts[:, :n].contiguous(); new_xyz = xyz[:, :m].contiguous(); nr = len(rk)
        radii = (ctypes.c_float * nr)(*[r for r, _ in rk]); ks = (ctypes.c_int * nr)(*[k for _, k in rk])
        res = {}
        for cpw in (2, 1, 0):
            lib.captra_ball_query_set_cpw(ctypes.c_int(cpw))
            outs = [torch.zeros(B, m, k, dtype=torch.int32, device=dev) for _, k in rk]
            ptrs = (ctypes.c_void_p * nr)(*[o.data_ptr() for o in outs])
            run = lambda: _lib.call("captra_ball_query_multi", B, n, m, nr, ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(ks, ctypes.c_void_p), new_xyz.data_ptr(), xyz.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p))
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res[cpw] = (1e3 * e0.elapsed_time(e1) / 20, outs)
        same = all(torch.equal(a, b) for a, b in zip(res[1][1], res[2][1]))
        print(f"B={B:3d} n={n} m={m} radii={nr}: two per wave {res[2][0]:6.1f} us, one per wave {res[1][0]:6.1f} us, dispatcher {res[0][0]:6.1f} us, identical lists {same}", flush=True)
lib.captra_ball_query_set_cpw(ctypes.c_int(0))
