"""HIP-graph capture of one tracking step.

A step is ~200 kernel launches, many of them tiny (the pose algebra after the networks); replaying
them as one hipGraph removes the launch gaps and the Python overhead from the frame loop.  Shapes are
static in tracking (B trajectories x N points), so the step is captured once per model and replayed
with the frame's cloud and the previous pose copied into static input buffers.

The reference has no equivalent (eager PyTorch, one ATen launch at a time, model.py:408-478).
"""
from __future__ import annotations

import os

import torch


# The streams the lanes of the re-crop loop run on: lane l's MAIN stream and the SIDE stream its RotationNet branch runs on,
# created ONCE per process and device, back to back.  Which hardware queue a HIP stream feeds is decided when it is created
# (GPU_MAX_HW_QUEUES = 4 queues), and two chains of kernels overlap only from different queues.  Round 2 captured a lane's step
# as ONE graph with the two networks as parallel branches: the runtime then runs the second branch on a stream of its own,
# made when the graph is instantiated -- wherever the queue assignment stands at that moment.  With the re-crop loop's four
# concurrent chains (2 lanes x 2 networks, a lane's sampler in front of its networks) every model object drew its own
# placement: 8.1 ms per 32-trajectory step, or 11.0, or 13.4 (tools/bench_otf.py --objects 14: 5 / 6 / 3 of 14 objects).
# SPLIT (TrackStepGraph(split_side=stream)): the step is captured as four LINEAR graphs (prep | RotationNet | CoordinateNet |
# read-out + pose fit) replayed on explicit streams with two events, so no stream is created behind the scenes and the four
# chains keep their queues for the life of the process: 8.56-8.64 ms for every one of 12 objects.  Measured alternatives:
# GPU_MAX_HW_QUEUES=8 with the split form 13.9 ms; a three-graph form whose side chain repeats the geometry kernels instead of
# waiting for a prep graph 8.5-8.7 ms and -6 % on the pre-cropped bench.  The pre-cropped lanes (TrackLanes) keep the one-graph
# form: in a fresh process it has been in its good placement in every session, and the split form's blocks vary more there
# (5.60-5.90 ms per step against 5.59-5.64; medians equal).
SPLIT_OTF_LANES = os.environ.get("CAPTRA_SPLIT_LANES", "1") != "0"      # (the environment switch is for A/B runs)
_LANE_STREAMS: dict = {}


def lane_streams(dev, lanes: int = 2):
    """[(main, side)] * lanes for device `dev`: process-wide, created back to back on first use."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    have = _LANE_STREAMS.setdefault(key, [])
    while len(have) < lanes:
        have.append((torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)))
    return have[:lanes]


class TrackStepGraph:
    def __init__(self, model, points: torch.Tensor, points_mean: torch.Tensor, pose: dict, labels: torch.Tensor | None = None,
                 warmup: int = 2, split_side=None, allow_split_k: bool = True):
        """model: EvalTrackModel (eval mode, on the GPU); points (B,3,N), points_mean (B,3,1), pose: example inputs.
        split_side: a stream -> the step is captured as four linear graphs and its RotationNet branch replays on that stream
        (see SPLIT_OTF_LANES); None -> one graph, the networks as its two branches when model.overlap_nets."""
        self.model = model
        self.split_side = split_side
        # a LANE of a larger batch (TrackLanes) must compute what the whole batch computes: the few-trajectory split-k rule of
        # EvalTrackModel._track_step goes by the batch it sees, so it is switched off for sub-batches
        self.allow_split_k = allow_split_k
        dev = points.device
        self.points = points.clone()
        self.points_mean = points_mean.clone()
        self.pose = {k: v.clone() for k, v in pose.items()}
        self.labels = None if labels is None else labels.clone()
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(warmup):          # folds weights, sets kernel attributes, warms the allocator
                self._step()
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        # the captured kernels read the folded weights through raw device pointers: own them (a module that drops its
        # cache must not hand the memory back to the allocator under the graph) and remember the version they belong to
        from .fold import collect_folded, weights_version
        self.weights = collect_folded(model)
        self.weights_version = weights_version()
        if split_side is not None and self._capture_split():
            return
        self.split_side = None
        try:
            self._capture()
        except RuntimeError:
            # the step forks CoordinateNet / RotationNet onto two streams (model.overlap_nets); should a runtime refuse
            # to capture the fork, capture the one-stream step instead
            if not getattr(model, "overlap_nets", False):
                raise
            model.overlap_nets = False
            torch.cuda.synchronize(dev)
            self._capture()

    def _capture(self):
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other host threads (RCCL's watchdog under torch.distributed) may touch the runtime while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.npcs_pred, self.out_pose = self._step()

    def _capture_split(self) -> bool:
        """[prep] -> [rot || coord] -> [post] as four linear graphs.  RotationNet's graph allocates from a pool of its own: it
        replays beside CoordinateNet's, so memory one of them frees while being captured must not be handed to the other."""
        from . import fused
        m = self.model
        inp = {"points": self.points, "points_mean": self.points_mean, "meta": {}}
        npcs_in = {"points": self.points, "points_mean": self.points_mean}
        if self.labels is not None:
            inp["labels"] = npcs_in["labels"] = self.labels
        if not m._overlap_nets(inp):
            return False
        cap = torch.cuda.Stream(device=self.points.device)
        pool_main, pool_side = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        self._graphs = [torch.cuda.CUDAGraph() for _ in range(4)]
        state = {}

        with fused.use_mlp_dtype(m.mlp_dtype):
            few = fused.split_k_rule(len(self.points), allow_few=self.allow_split_k)     # (as EvalTrackModel._track_step)

        def capture(g, pool, fn):
            with torch.cuda.graph(g, pool=pool, stream=cap, capture_error_mode="thread_local"), torch.no_grad(), fused.use_mlp_dtype(m.mlp_dtype), fused.split_k(few):
                return fn()

        def prep():
            m._step_begin(inp, npcs_in, self.pose)
            return m._step_prep(inp, npcs_in, self.pose)

        if not capture(self._graphs[0], pool_main, prep):
            return False
        state["raw"] = capture(self._graphs[1], pool_side, lambda: m._step_rot(inp, npcs_in, self.pose))
        self.npcs_pred = capture(self._graphs[2], pool_main, lambda: m._step_coord(npcs_in))

        def post():
            inp["_raw"] = state["raw"]
            return m._step_post(inp, npcs_in, self.npcs_pred, self.pose)

        self.out_pose = capture(self._graphs[3], pool_main, post)
        self._keep = (inp, npcs_in, state)                 # tensors one graph hands to the next
        self._ev_prep, self._ev_rot = torch.cuda.Event(), torch.cuda.Event()
        return True

    def _replay_graphs(self):
        if self.split_side is None:
            self.graph.replay()
            return
        main, side = torch.cuda.current_stream(self.points.device), self.split_side
        g_prep, g_rot, g_coord, g_post = self._graphs
        g_prep.replay()
        self._ev_prep.record(main)
        side.wait_event(self._ev_prep)
        with torch.cuda.stream(side):
            g_rot.replay()
            self._ev_rot.record(side)
        g_coord.replay()
        main.wait_event(self._ev_rot)
        g_post.replay()

    def _step(self):
        input = {"points": self.points, "points_mean": self.points_mean, "meta": {}}
        npcs_input = {"points": self.points, "points_mean": self.points_mean}
        if self.labels is not None:
            input["labels"] = self.labels
            npcs_input["labels"] = self.labels
        prev = getattr(self.model, "_no_split_k", False)
        self.model._no_split_k = prev or not self.allow_split_k
        try:
            return self.model.track_step(input, npcs_input, self.pose)
        finally:
            self.model._no_split_k = prev

    def stale(self) -> bool:
        """True when a module UNDER THIS GRAPH'S MODEL re-folded (or dropped) its weights after the capture: the replay would
        still compute with the weights the graph owns, i.e. the OLD parameters.  The process-wide version counter
        (fold.weights_version) is only the fast path: when it moved -- any model of the process may have re-folded -- the
        model's current packed layers are compared by identity with the ones captured, and an unrelated model's change
        leaves this graph valid."""
        from .fold import collect_folded, weights_version
        now = weights_version()
        if now == self.weights_version:
            return False
        current = collect_folded(self.model)
        if len(current) == len(self.weights) and all(a is b for a, b in zip(current, self.weights)):
            self.weights_version = now
            return False
        return True

    def replay(self, points, points_mean, pose, labels=None):
        """Copies the inputs into the captured buffers, replays, returns the (static) output pose dict —
        clone it if it must survive the next replay."""
        if self.stale():
            raise RuntimeError("the model's weights changed after this hipGraph was captured (train()/load_state_dict()/.to()); "
                               "capture a new TrackStepGraph")
        pairs = [(points, self.points), (points_mean, self.points_mean)]
        if pose is not self.pose:          # (a lane of TrackLanes hands its pose over in place)
            pairs += [(pose[k], self.pose[k]) for k in self.pose]
        if self.labels is not None and labels is not None:
            pairs.append((labels, self.labels))
        # the step's inputs into the captured buffers: one launch when they are plain device tensors of the captured types
        if all(a.is_cuda and a.is_contiguous() and a.dtype == b.dtype and a.shape == b.shape and a.element_size() % 4 == 0 for a, b in pairs):
            from . import fused
            fused.copy_multi(pairs)
        else:
            for a, b in pairs:
                b.copy_(a)
        self._replay_graphs()
        return self.out_pose


class TrackLanes:
    """The B trajectories of a rank as `lanes` independent sub-batches, each with its own captured step (TrackStepGraph) and
    its own stream, FREE-RUNNING: a lane hands its pose over to itself and starts its next frame without waiting for the
    others, so the lanes drift apart and one lane's furthest-point sampling (one workgroup per cloud: B of 256 CUs busy)
    runs under another lane's MFMA kernels.  Trajectories never exchange anything (frame i of a trajectory needs its own
    pose i-1 only, model.py:422,454-461), so the split changes no result; what a consumer needs — the whole batch's poses
    of a frame, for the all-gather or the result list — is assembled in a small ring of (B, ...) records that the lanes
    write their slices of and `gather` hands out on the caller's stream (GPU-side event waits, the lanes do not stop).
    Measured: 2 lanes +3.5 % frames/s at B = 32; joining the lanes every frame instead gives the gain back (-0.5 %)."""

    def __init__(self, model, points: torch.Tensor, points_mean: torch.Tensor, pose: dict, lanes: int = 2, ring: int = 4,
                 keep_npcs: bool = False):
        """keep_npcs: also keep CoordinateNet's per-point outputs of every frame in the ring (`gather(slot, npcs=True)`; the
        track loop's pred_dict['npcs_pred'], model.py:476-478)."""
        B = points.shape[0]
        if lanes < 1 or B % lanes:
            raise ValueError(f"{B} trajectories do not split into {lanes} lanes")
        dev = points.device
        per = B // lanes
        self.slices = [slice(l * per, (l + 1) * per) for l in range(lanes)]
        self.graphs = [TrackStepGraph(model, points[s].contiguous(), points_mean[s].contiguous(), {k: v[s].contiguous() for k, v in pose.items()}, allow_split_k=False)
                       for s in self.slices]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        self.ring = [{k: torch.empty_like(v) for k, v in pose.items()} for _ in range(ring)]
        self.npcs_ring = None
        if keep_npcs:
            first = {k: v for k, v in self.graphs[0].npcs_pred.items() if torch.is_tensor(v)}
            self.npcs_ring = [{k: v.new_empty((B,) + tuple(v.shape[1:])) for k, v in first.items()} for _ in range(ring)]
        self.written = [[torch.cuda.Event() for _ in range(lanes)] for _ in range(ring)]
        self.consumed = [None] * ring          # recorded on the consumer's stream when the slot was handed out and read
        self.frame = 0
        self._pending = None                   # (slot, stream) handed out by the last gather, not yet marked consumed
        self.set_pose(pose)

    def stale(self) -> bool:
        return any(g.stale() for g in self.graphs)

    def set_pose(self, pose: dict) -> None:
        """(Re)start the trajectories from `pose` (B-major dict) — the initial pose of the track loop (model.py:394)."""
        cur = torch.cuda.current_stream()
        for g, s, st in zip(self.graphs, self.slices, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for k in g.pose:
                    g.pose[k].copy_(pose[k][s])

    def step(self, points: torch.Tensor, points_mean: torch.Tensor, labels=None, sync_inputs: bool = False) -> int:
        """Enqueue one frame on every lane; returns the ring slot its poses will be in.  The frame's inputs must already be
        resident; if the caller's stream is still producing them pass sync_inputs=True (the lanes then wait for that
        stream, which joins them whenever it also carries the previous frame's gather)."""
        from . import fused
        self._mark_consumed()
        slot = self.frame % len(self.ring)
        self.frame += 1
        cur = torch.cuda.current_stream()
        for l, (g, s, st) in enumerate(zip(self.graphs, self.slices, self.streams)):
            if self.frame == 1 or sync_inputs:
                st.wait_stream(cur)
            if self.consumed[slot] is not None:
                st.wait_event(self.consumed[slot])
            with torch.cuda.stream(st):
                out = g.replay(points[s], points_mean[s], g.pose, None if labels is None else labels[s])
                # the frame's record and the hand-over inside the lane (and CoordinateNet's maps): ONE launch, not ten copies
                pairs = [(out[k], self.ring[slot][k][s]) for k in out] + [(out[k], g.pose[k]) for k in out]
                if self.npcs_ring is not None:
                    pairs += [(g.npcs_pred[k], dst[s]) for k, dst in self.npcs_ring[slot].items()]
                fused.copy_multi(pairs)
                self.written[slot][l].record(st)
        return slot

    def gather(self, slot: int, npcs: bool = False):
        """The frame's poses of all B trajectories (npcs=True: and CoordinateNet's outputs), valid on the current stream
        (which waits for the lanes' writes on the GPU; the host does not block).  The dicts are ring records: they are
        overwritten `ring` frames later."""
        cur = torch.cuda.current_stream()
        for ev in self.written[slot]:
            cur.wait_event(ev)
        self._pending = (slot, cur)
        return (self.ring[slot], self.npcs_ring[slot]) if npcs else self.ring[slot]

    def _mark_consumed(self):
        if self._pending is not None:
            slot, st = self._pending
            ev = self.consumed[slot] or torch.cuda.Event()
            ev.record(st)
            self.consumed[slot] = ev
            self._pending = None


class BackbonePipe:
    """`PointNet2Msg.forward` over a STREAM of independent batches, as a two-stage pipeline on two streams: everything of a
    batch that depends on the coordinates only -- both furthest-point samplings, both ball queries, the 3-NN weights
    (`precompute_geometry`, backbones.py:30-69 of the reference run them inside forward) -- is one captured graph on the
    geometry stream, the shared MLPs / feature propagation (`forward(geom=...)`) another on the MLP stream, and batch
    t + 1's geometry runs beside batch t's MLPs.  The samplers are one workgroup per cloud (8 clouds of BASELINE.json
    configs[4]: 8 of 256 CUs for 1.9 of the step's 4.6 ms), the MLP kernels fill the chip: side by side a batch costs
    max(geometry, MLPs) instead of their sum.  Batches are independent (no pose feedback in the backbone), so every output
    is the one `net(x)` returns, bit for bit.  `depth` slots of static buffers; `push` enqueues a batch and returns its slot,
    `output(slot)` makes the caller's stream wait for it.

    Not for the tracking step: there frame t + 1's cloud is canonicalised with pose t (model.py:422,454-461)."""

    def __init__(self, net, x: torch.Tensor, depth: int = 2, warmup: int = 2, reserve: int | None = None):
        if depth < 2:
            raise ValueError("a pipeline needs at least two slots")
        if net.training or not x.is_cuda:
            raise ValueError("BackbonePipe: eval-mode network on the GPU")
        dev = x.device
        self.net, self.depth, self.t = net, depth, 0
        # The MLP graph's PERSISTENT kernels (the fp32 SA scales: one workgroup per CU slot, centres walked statically) are sized
        # for `reserve` CUs fewer (default: one per cloud; A/B: CAPTRA_PIPE_RESERVE): a sampler workgroup of the other stream holds
        # its CU for the whole 1.5 ms of its rounds, and a persistent workgroup that finds its CU taken starts when another one
        # ENDS -- the launch takes twice as long (configs[4]: the pipelined batch 3.23 ms against max(2.30, 2.05); a
        # high-priority geometry stream does not change that, a CU-masked MLP stream makes it worse: 3.86 ms, the masked
        # kernels run 2.05 -> 3.38 ms because every grid sized for 256 CUs then takes a second round).
        # What the MLP graph does instead (CAPTRA_PIPE_DYNAMIC, default on): its SA kernels hand their centres out through a
        # counter (captra_launch_opts::dyn_slot), so a workgroup that becomes resident late finds the work done; `reserve` then
        # defaults to 0.
        B, _, N = x.shape
        self.dynamic = os.environ.get("CAPTRA_PIPE_DYNAMIC", "1") != "0"
        self.reserve = int(os.environ.get("CAPTRA_PIPE_RESERVE", str(reserve if reserve is not None else (0 if self.dynamic else B))))
        self._dyn_pool = torch.zeros(depth, 64, dtype=torch.int32, device=dev)
        self.geom_stream = torch.cuda.Stream(device=dev)
        self.mlp_stream = torch.cuda.Stream(device=dev)
        self.x = [x.clone() for _ in range(depth)]
        self.x_n3 = [torch.empty(B, N, 3, dtype=x.dtype, device=dev) for _ in range(depth)]
        cur = torch.cuda.current_stream(dev)
        self.mlp_stream.wait_stream(cur)
        with torch.cuda.stream(self.mlp_stream), torch.no_grad():
            for _ in range(warmup):                    # folds weights, sets kernel attributes, warms the allocator
                self._geometry(0)
                net(self.x[0], input_n3=self.x_n3[0], geom=self._geometry(0))
        cur.wait_stream(self.mlp_stream)
        torch.cuda.synchronize(dev)
        if self._geometry(0) is None:
            raise ValueError("BackbonePipe: the fused samplers do not cover this cloud size")
        from .fold import collect_folded, weights_version
        self.weights = collect_folded(net)             # the captured kernels read the folded weights through raw pointers
        self.weights_version = weights_version()
        self.g_geom, self.g_mlp, self.geom, self.out = [], [], [], []
        import ctypes
        from . import _lib
        cap = torch.cuda.Stream(device=dev)
        for s in range(depth):
            gg, gm = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, stream=cap, capture_error_mode="thread_local"), torch.no_grad():
                geom = self._geometry(s)
            # (per-call options of the launches captured here: captra_launch_opts::reserved_cus / dyn_slot -- a slot of this graph's
            # own pool per launch, round robin; the C library remembers nothing)
            with _lib.launch_options(reserved_cus=self.reserve, dyn_pool=(self._dyn_pool[s].data_ptr(), 64) if self.dynamic else None):
                with torch.cuda.graph(gm, stream=cap, capture_error_mode="thread_local"), torch.no_grad():
                    out = net(self.x[s], input_n3=self.x_n3[s], geom=geom)
            self.g_geom.append(gg); self.g_mlp.append(gm); self.geom.append(geom); self.out.append(out)
        self.geom_ready = [torch.cuda.Event() for _ in range(depth)]
        self.mlp_done = [None] * depth
        self.consumed = [None] * depth                 # recorded on the consumer's stream at the next push after `output(slot)`
        self._pending = []                             # (slot, stream) handed out since the last push

    def _geometry(self, s: int):
        x = self.x[s]
        xyz = x[:, :3] if x.shape[1] > 3 else x
        self.x_n3[s].copy_(xyz.transpose(1, 2))
        return self.net.precompute_geometry(self.x_n3[s])

    def push(self, x: torch.Tensor | None = None) -> int:
        """Enqueue one batch (x (B,C,N) as captured; None: the slot's resident input again).  Returns the slot."""
        for slot, st in self._pending:                 # whoever took a slot's output has enqueued its reads by now
            ev = self.consumed[slot] or torch.cuda.Event()
            ev.record(st)
            self.consumed[slot] = ev
        self._pending = []
        s = self.t % self.depth
        self.t += 1
        gs, ms = self.geom_stream, self.mlp_stream
        if self.mlp_done[s] is not None:               # the slot's previous batch still reads its input / geometry
            gs.wait_event(self.mlp_done[s])
        if self.consumed[s] is not None:               # ... and its consumer the output (geometry first: the MLP graph follows it)
            gs.wait_event(self.consumed[s])
        if x is not None:
            gs.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(gs):
            if x is not None:
                self.x[s].copy_(x)
            self.g_geom[s].replay()
            self.geom_ready[s].record(gs)
        ms.wait_event(self.geom_ready[s])
        with torch.cuda.stream(ms):
            self.g_mlp[s].replay()
            ev = self.mlp_done[s] or torch.cuda.Event()
            ev.record(ms)
            self.mlp_done[s] = ev
        return s

    def output(self, slot: int) -> torch.Tensor:
        """The slot's output (B,out_dim,N), valid on the current stream; overwritten `depth` pushes later."""
        cur = torch.cuda.current_stream(self.out[slot].device)
        if self.mlp_done[slot] is not None:
            cur.wait_event(self.mlp_done[slot])
        self._pending.append((slot, cur))
        return self.out[slot]

    def drain(self) -> None:
        """The caller's stream waits for everything pushed so far."""
        cur = torch.cuda.current_stream(self.x[0].device)
        cur.wait_stream(self.geom_stream)
        cur.wait_stream(self.mlp_stream)
