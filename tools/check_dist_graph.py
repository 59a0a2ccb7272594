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
ex = PoseExchange(8, cfg["num_parts"], device, 1, 0, collective=True)       # world 1, but THROUGH the communicator
for i in range(5):
    pose = graph.replay(f1["points"], f1["points_mean"], pose)
    ex.wait()
    ex.all_gather(pose, async_op=True)                                       # bench.py's per-step exchange
    dist.barrier()
ex.wait()
torch.cuda.synchronize()
assert all(torch.isfinite(v).all() for v in pose.values())
assert torch.equal(ex.gathered, ex.local)
print("ok: captured with overlap_nets =", model.overlap_nets)
# bench.py's default from 32 trajectories per GPU on: two free-running lanes, the all-gather fed from the ring records
lanes = TrackLanes(model, f1["points"], f1["points_mean"], {k: v.clone() for k, v in model.feed_dict[0]["gt_part"].items()}, lanes=2)
for i in range(12):
    rec = lanes.gather(lanes.step(f1["points"], f1["points_mean"]))
    ex.wait()
    ex.all_gather(rec, async_op=True)
ex.wait()
dist.barrier()
torch.cuda.synchronize()
assert torch.isfinite(ex.gathered).all() and torch.equal(ex.gathered, ex.local)
print("ok: free-running lanes + all-gather")
# the product harness's exchange (captra_amd.track.FramePoseGather: async all-gather per frame, waited for at the end)
from captra_amd.track import FramePoseGather, Ranks  # noqa: E402
ranks = Ranks()
ranks.dist = dist                         # the group initialised above (Ranks.init skips it at world 1)
assert ranks.world == 1
fg = FramePoseGather(8, cfg["num_parts"], device, ranks, 8)
fg.ex.collective = True                   # world 1, but through the communicator
for i in range(3):
    fg(i, pose)
recs = fg.finish(4)                       # one more frame than this rank had: the short-batch path (invalid records)
assert len(recs) == 4 and all(torch.isfinite(r).all() for r in recs) and float(recs[3][..., 13].max()) == 0.0
print("ok: track harness frame exchange")
dist.barrier()
dist.destroy_process_group()
print("ok: process group destroyed")
