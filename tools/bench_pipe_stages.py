"""Stage times of captra_amd.graph.BackbonePipe on BASELINE.json configs[4] (8 clouds of 16384 points): the geometry graph alone, the MLP
graph alone, both pipelined, and each stage's time INSIDE the pipeline (events around every replay on its own stream).
    python tools/bench_pipe_stages.py [clouds]        CAPTRA_PIPE_DYNAMIC=0 / CAPTRA_PIPE_RESERVE=n: the A/B switches of DESIGN.md 3.4"""
import copy, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from captra_amd import synthetic as clouds
from captra_amd.synthetic import make_state_dict
from captra_amd.backbones import PointNet2Msg
from captra_amd.configs import make_config
from captra_amd.graph import BackbonePipe
dev = torch.device("cuda:0")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 16384
x = torch.from_numpy(np.ascontiguousarray(np.stack([clouds.s_uni(i, N) for i in range(B)]).astype(np.float32).transpose(0, 2, 1))).to(dev)
cfg = copy.deepcopy(make_config("1"))
cfg["pointnet"]["camera"]["sa1"]["npoint"] = 2048
cfg["pointnet"]["camera"]["sa2"]["npoint"] = 512
net = PointNet2Msg(cfg, 128)
net.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=21))
net = net.to(dev).eval()
pipe = BackbonePipe(net, x)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
with torch.cuda.stream(pipe.geom_stream):
    print("geometry graph alone  %.3f ms" % t(pipe.g_geom[0].replay))
with torch.cuda.stream(pipe.mlp_stream):
    print("MLP graph alone       %.3f ms" % t(pipe.g_mlp[0].replay))
print("pipelined             %.3f ms" % t(pipe.push, 40))
# stage times INSIDE the pipelined run: events around each graph replay on its own stream
n = 30
ge = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
me = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
torch.cuda.synchronize()
for i in range(n):
    s = pipe.t % pipe.depth
    pipe.t += 1
    gs, ms = pipe.geom_stream, pipe.mlp_stream
    if pipe.mlp_done[s] is not None:
        gs.wait_event(pipe.mlp_done[s])
    with torch.cuda.stream(gs):
        ge[i][0].record(gs); pipe.g_geom[s].replay(); ge[i][1].record(gs); pipe.geom_ready[s].record(gs)
    ms.wait_event(pipe.geom_ready[s])
    with torch.cuda.stream(ms):
        me[i][0].record(ms); pipe.g_mlp[s].replay(); me[i][1].record(ms); pipe.mlp_done[s].record(ms)
torch.cuda.synchronize()
g = sorted(a.elapsed_time(b) for a, b in ge[5:]); m = sorted(a.elapsed_time(b) for a, b in me[5:])
print("inside the pipeline: geometry stage median %.3f ms, MLP stage median %.3f ms" % (g[len(g) // 2], m[len(m) // 2]))
print("geometry start-to-start %.3f ms" % (ge[5][0].elapsed_time(ge[-1][0]) / (n - 6)))
