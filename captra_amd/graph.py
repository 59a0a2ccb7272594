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

    def replay(self, points, points_mean, pose, labels=None):
        """Copies the inputs into the captured buffers, replays, returns the (static) output pose dict —
        clone it if it must survive the next replay."""
        self.points.copy_(points)
        self.points_mean.copy_(points_mean)
        for k in self.pose:
            self.pose[k].copy_(pose[k])
        if self.labels is not None and labels is not None:
            self.labels.copy_(labels)
        self.graph.replay()
        return self.out_pose
