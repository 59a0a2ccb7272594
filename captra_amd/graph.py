"""HIP-graph capture of one tracking step.

A step is ~200 kernel launches, many of them tiny (the pose algebra after the networks); replaying
them as one hipGraph removes the launch gaps and the Python overhead from the frame loop.  Shapes are
static in tracking (B trajectories x N points), so the step is captured once per model and replayed
with the frame's cloud and the previous pose copied into static input buffers.

The reference has no equivalent (eager PyTorch, one ATen launch at a time, model.py:408-478).
"""
from __future__ import annotations

import torch


class TrackStepGraph:
    def __init__(self, model, points: torch.Tensor, points_mean: torch.Tensor, pose: dict, labels: torch.Tensor | None = None,
                 warmup: int = 2):
        """model: EvalTrackModel (eval mode, on the GPU); points (B,3,N), points_mean (B,3,1), pose: example inputs."""
        self.model = model
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

    def _step(self):
        input = {"points": self.points, "points_mean": self.points_mean, "meta": {}}
        npcs_input = {"points": self.points, "points_mean": self.points_mean}
        if self.labels is not None:
            input["labels"] = self.labels
            npcs_input["labels"] = self.labels
        return self.model.track_step(input, npcs_input, self.pose)

    def stale(self) -> bool:
        """True when some module re-folded (or dropped) its weights after this graph was captured: the replay would still
        compute with the weights it owns, i.e. the OLD parameters."""
        from .fold import weights_version
        return weights_version() != self.weights_version

    def replay(self, points, points_mean, pose, labels=None):
        """Copies the inputs into the captured buffers, replays, returns the (static) output pose dict —
        clone it if it must survive the next replay."""
        if self.stale():
            raise RuntimeError("the model's weights changed after this hipGraph was captured (train()/load_state_dict()/.to()); "
                               "capture a new TrackStepGraph")
        self.points.copy_(points)
        self.points_mean.copy_(points_mean)
        if pose is not self.pose:          # (a lane of TrackLanes hands its pose over in place)
            for k in self.pose:
                self.pose[k].copy_(pose[k])
        if self.labels is not None and labels is not None:
            self.labels.copy_(labels)
        self.graph.replay()
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
        self.graphs = [TrackStepGraph(model, points[s].contiguous(), points_mean[s].contiguous(), {k: v[s].contiguous() for k, v in pose.items()})
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
                for k in out:
                    self.ring[slot][k][s].copy_(out[k])
                    g.pose[k].copy_(out[k])    # hand-over inside the lane
                if self.npcs_ring is not None:
                    for k, dst in self.npcs_ring[slot].items():
                        dst[s].copy_(g.npcs_pred[k])
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
