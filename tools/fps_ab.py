import sys, ctypes, time, numpy as np, torch
sys.path.insert(0, '.')
from captra_amd import _lib, fused
from captra_amd import synthetic as clouds
dev = torch.device('cuda:0')
lib = _lib.lib()
for B in (1, 16, 32):
    x = torch.from_numpy(np.stack([clouds.s_nocs(1000 + i)[0] for i in range(B)])).to(dev).contiguous()
    for n, m in ((4096, 512), (512, 128)):
        xx = x[:, :n].contiguous()
        for defer in (0, 1, 0, 1):
            lib.captra_fps_set_defer(ctypes.c_int(defer))
            for _ in range(3): r = fused.fps_gather(xx, m)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): r = fused.fps_gather(xx, m)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"B={B} n={n} m={m} defer={defer}: {dt*1e6:8.1f} us  {dt*1e6/(m-1):.3f} us/round", flush=True)
