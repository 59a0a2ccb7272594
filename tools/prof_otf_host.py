import sys, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '/root/repo')
from captra_amd import nocs_otf
from captra_amd.synthetic import make_frame
dev = torch.device('cuda:0')
items = []
for b in range(32):
    depth, mask, center, pose = make_frame(1 + b % 3)
    items.append((torch.from_numpy(depth.astype(np.int32)).to(dev), torch.from_numpy(mask).to(dev), center, 0.3, pose))
for _ in range(3): nocs_otf.full_data_batch(items, 4096)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): nocs_otf.full_data_batch(items, 4096)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
