"""Graph capture + replay of the tracking step inside an initialised RCCL process group (world size 1 on one GPU): the
same code path bench.py takes per rank under torch.distributed.run, minus the peers."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import bench  # noqa: E402
from captra_amd.graph import TrackLanes, TrackStepGraph  # noqa: E402
from captra_amd.parallel import PoseExchange  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
warm = torch.ones(4, device=device)
dist.all_reduce(warm)                       # communicator + watchdog up before the capture
cfg, sd, model, data = bench.build_workload(8, device)
pose = {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}
f1 = model.feed_dict[1]
graph = TrackStepGraph(model, f1["points"], f1["points_mean"], pose)
ex = PoseExchange(8, cfg["num_parts"], device, 1, 0)
for i in range(5):
    pose = graph.replay(f1["points"], f1["points_mean"], pose)
    import torch.distributed as d2
    d2.all_gather_into_tensor(ex.gathered, ex.local.copy_(__import__("captra_amd.parallel", fromlist=["pack_pose"]).pack_pose(pose)))
    dist.barrier()
torch.cuda.synchronize()
assert all(torch.isfinite(v).all() for v in pose.values())
print("ok: captured with overlap_nets =", model.overlap_nets)
# bench.py's default from 32 trajectories per GPU on: two free-running lanes, the all-gather fed from the ring records
from captra_amd.parallel import pack_pose  # noqa: E402
lanes = TrackLanes(model, f1["points"], f1["points_mean"], {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}, lanes=2)
for i in range(12):
    rec = lanes.gather(lanes.step(f1["points"], f1["points_mean"]))
    dist.all_gather_into_tensor(ex.gathered, ex.local.copy_(pack_pose(rec)))
dist.barrier()
torch.cuda.synchronize()
assert torch.isfinite(ex.gathered).all()
print("ok: free-running lanes + all-gather")
dist.destroy_process_group()
