import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from captra_amd import fused
dev = torch.device('cuda:0')
cfg, sd, model, data = bench.build_workload(8, dev)
pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
outs = {}
for dt in ("fp32", "bf16"):
    fused.set_mlp_dtype(dt)
    p = {k: v.clone() for k, v in pose.items()}
    with torch.no_grad():
        npcs, p1 = model.track_step(dict(model.feed_dict[1]), dict(model.npcs_feed_dict[1]), p)
    outs[dt] = (npcs["nocs"].cpu().numpy(), npcs["seg"].cpu().numpy(), {k: v.cpu().numpy() for k, v in p1.items()})
fused.set_mlp_dtype("fp32")
a, b = outs["fp32"], outs["bf16"]
print("nocs  max|diff| %.4g  mean %.4g" % (np.abs(a[0] - b[0]).max(), np.abs(a[0] - b[0]).mean()))
print("seg   max|diff| %.4g" % np.abs(a[1] - b[1]).max(), " label flips:", int((a[1].argmax(1) != b[1].argmax(1)).sum()), "of", a[1].shape[0] * a[1].shape[2])
R0, R1 = a[2]["rotation"], b[2]["rotation"]
ang = np.degrees(np.arccos(np.clip((np.einsum("bpij,bpij->bp", R0, R1) - 1) / 2, -1, 1)))
print("rotation angle between fp32 and bf16 poses (deg):", ang.ravel().round(3))
print("translation diff (m):", np.abs(a[2]["translation"] - b[2]["translation"]).max(), " scale diff:", np.abs(a[2]["scale"] - b[2]["scale"]).max())
