import sys, ctypes, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from captra_amd import _lib, fused
from captra_amd import synthetic as clouds
dev = torch.device('cuda:0')
def surf(seed, n):
    rng = np.random.default_rng(seed)
    th, h = rng.random(n) * 2 * np.pi, rng.random(n) - 0.5
    pts = np.stack([0.2 * np.cos(th), h, 0.2 * np.sin(th)], -1) + rng.normal(0, 0.004, (n, 3))
    pts[: n // 5] = rng.random((n // 5, 3)) - 0.5
    return rng.permutation(pts).astype(np.float32)
stats = torch.zeros(8, dtype=torch.int64, device=dev)
for n, m, B in [(15000, 4096, 32), (20480, 4096, 32), (20480, 2048, 8), (16384, 4096, 32), (16384, 2048, 8), (12288, 2048, 8), (8192, 1024, 8)]:
    for kind in ("surface", "uniform"):
        xyz = np.stack([surf(i, n) if kind == "surface" else clouds.s_uni(i, n) for i in range(B)])
        x = torch.from_numpy(xyz).to(dev)
        for pm in (0, 4096):
            _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(pm))
            NW = 8
            for _ in range(2): r = fused.fps_gather(x, m)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): r = fused.fps_gather(x, m)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            extra = ""
            if pm:
                stats.zero_()
                _lib.lib().captra_fps_set_stats(ctypes.c_void_p(stats.data_ptr()))
                fused.fps_gather(x, m); torch.cuda.synchronize()
                _lib.lib().captra_fps_set_stats(ctypes.c_void_p(0))
                u, r, *ph = stats.tolist()[:6]; rounds = stats[6].item() / B / NW
                per = [p / B / NW / (m - 1) for p in ph]
                extra = (f"  bucket updates/round/wave {u/B/NW/(m-1):5.2f}  refreshes {r/B/NW/(m-1):5.2f}  (unpruned: {n/512:.0f})"
                         f"  rounds {rounds:.0f}  cycles/pick/wave: test+upd {per[0]:.0f} cand {per[1]:.0f} barrier {per[2]:.0f} exchange {per[3]:.0f}")
            print(f"n={n} m={m} B={B} {kind:8s} pruned_min={pm}: {dt*1e3:8.3f} ms  {dt*1e6/(m-1):6.3f} us/round{extra}", flush=True)
        _lib.lib().captra_fps_set_pruned_min(ctypes.c_int(8192))
