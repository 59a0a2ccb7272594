"""The tracking loop: `EvalTrackModel` (and its `BaseModel`) on the MI355X path.

Mirrors the contract of the reference's network/models/model.py: `BaseModel` (l.27-104) and
`EvalTrackModel` (l.311-600): `set_data(data)`, `test(save, no_eval, epoch)`, attributes
`pred_dict = {'poses': [pose]*T, 'npcs_pred': [None, {...}]*}` and `loss_dict`.
`data` is a list over frames of dicts
  {'points' (B,3,N), 'labels' (B,N), 'nocs' (B,3,N),
   'meta': {'path': [str]*B, 'nocs2camera': [{'rotation' (B,3,3), 'translation' (B,3,1), 'scale' (B,)}]*P,
            'points_mean' (B,3,1), 'nocs_corners' (B,P,2,3)}}.
Frame i consumes the pose predicted for frame i-1 (strictly sequential, model.py:408-478);
trajectories of one batch are independent, which is what shards over GPUs (parallel.py).

Out of scope here (SURVEY.md §8f "next"): the on-the-fly depth crop of `nocs_otf` (model.py:425-452,
needs cv2 + the dataset) and the IoU/segmentation losses of compute_loss (loss.py, bbox_utils.py).
"""
from __future__ import annotations

import pickle
from copy import deepcopy
from os.path import join as pjoin

import os

import numpy as np
import torch
import torch.nn as nn

from .networks import CoordNet, PartCanonNet
from .pose_utils.part_dof_utils import add_noise_to_part_dof, consume_noise_draws, eval_part_full, part_model_batch_to_part
from .utils import Timer, add_dict, cvt_torch, divide_dict, ensure_dirs, get_ith_from_batch


def write_result_pickles(experiment_dir: str, records) -> None:
    """records: [(file name, per-trajectory result dict)] -> <experiment_dir>/results/data/<instance>_<track>.pkl
    (reference model.py:503-509)."""
    save_path = pjoin(experiment_dir, "results", "data")
    ensure_dirs([save_path])
    for name, rec in records:
        with open(pjoin(save_path, name), "wb") as f:
            pickle.dump(rec, f)


class BaseModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_parts = int(cfg["num_parts"])
        self.num_joints = int(cfg["num_joints"])
        self.device = cfg["device"]
        self.network_type = cfg["network"]["type"]
        raw = cfg["pose_perturb"]
        self.pose_perturb_cfg = {"type": raw["type"], "scale": raw["s"], "translation": raw["t"],
                                 "rotation": float(np.deg2rad(raw["r"]))}
        self.sym = cfg["obj_sym"]
        self.cfg = cfg
        self.feed_dict = {}
        self.pred_dict = {}
        self.loss_dict = {}
        self.per_diff_dict = {}

    def record_per_diff(self, data, per_diff):
        for i, path in enumerate(data["meta"]["path"]):
            instance, track_num, frame_i = path.split(".")[-2].split("/")[-3:]
            self.per_diff_dict.setdefault(f"{instance}_{track_num}_{frame_i}", {}).update(get_ith_from_batch(per_diff, i))


OTF_FIRST_BOUND = 5                 # the first frame's stride bound, in units of N (5 N = the longest list the fast path takes)
OTF_DEFER = os.environ.get("CAPTRA_OTF_DEFER", "1") != "0"     # nocs_otf: no round trip for the crops' member counts either (A/B: 0 = the synchronous stage)


class _OtfCheck:
    """The deferred verdict of one frame's sync-free re-crop: the device word [rare-path instance met, longest candidate list]
    (captra_amd/nocs_otf.py) on its way to pinned host memory behind the crop launch.  `read()` -- called when the NEXT frame is
    about to be enqueued -- waits for that copy (work enqueued a frame ago: the host's only wait, and the GPU never waits for the
    host) and returns (rare, longest)."""
    _pinned: list = []

    def __init__(self, info):
        self.host = _OtfCheck._pinned.pop() if _OtfCheck._pinned else torch.empty(4, dtype=torch.int32).pin_memory()
        self.host.copy_(info, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def read(self):
        self.event.synchronize()
        rare, longest = int(self.host[0]), int(self.host[1])
        _OtfCheck._pinned.append(self.host)
        return bool(rare), longest


def _otf_bound(longest: int, n: int) -> int:
    """The sampler's padded stride for the next frame: the last frame's longest candidate list + 3 %, in steps of 1024, within
    [N, 5 N] (a list that outgrows it is a rare-path instance: the frame runs again on the synchronous stage).  Tight on purpose: the
    pruned sampler keeps a cloud's buckets in registers up to 16384 points and spills slots to LDS beyond (fps_pruned.hip), and a
    tracked object's crop changes by a few points per frame."""
    return int(min(5 * n, max(n, -(-(longest + longest // 32) // 1024) * 1024)))


OTF_POSE_ON_DEVICE = os.environ.get("CAPTRA_OTF_POSE_ON_DEVICE", "1") != "0"   # nocs_otf: the crop box from the device-resident pose (A/B: 0 = via the host)
_OTF_LANE_STREAMS: dict = {}      # device index -> the two lane streams of EvalTrackModel._forward_otf_lanes


class EvalTrackModel(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.net = PartCanonNet(cfg)
        self.npcs_net = CoordNet(cfg)
        self.tree = cfg["obj_tree"]
        self.root = [p for p in range(len(self.tree)) if self.tree[p] == -1][0]
        self.gt_init = cfg["init_frame"]["gt"]
        self.nocs_otf = bool(cfg.get("nocs_otf", False))
        self.radius = cfg["data_radius"]
        self.track_cfg = cfg["track_cfg"]
        self.npcs_feed_dict = []
        self.timer = Timer(True)
        self.time_dict = {"npcs_net": 0.0, "rot_all": 0.0}
        # single-part objects: RotationNet canonicalises with the very pose CoordNet used, so both nets see
        # the same cloud and FPS / ball query / 3-NN run once per frame instead of twice
        self.share_geometry = True
        self.overlap_geometry = os.environ.get("CAPTRA_OVERLAP_GEOM", "1") != "0"   # the shared geometry's two levels side by side
        self.overlap_nets = True     # CoordinateNet and RotationNet side by side on two streams (one part: they share the cloud)
        # STREAMED level-1 sampling (backbones.precompute_geometry_streamed): the 4096 -> 512 sampler as this many launches on a
        # stream of its own, the networks' first level walking the centres as they are picked; 0 / 1 = the sampler, then the
        # networks.  Same picks, same neighbour lists, same bits.
        self.sampler_chunks = int(cfg.get("sampler_chunks", os.environ.get("CAPTRA_SAMPLER_CHUNKS", "0")))
        # bf16 mode, one part: the first level of BOTH networks inside the sampler's launch (csrc/sa_bf16.hip level-1 stream kernel:
        # sampler workgroups publish their picks, the other workgroups run ball query + shared MLPs of the published centres).
        # Same picks, lists and features, bit for bit (tests/test_l1_stream_gpu.py); cfg['l1_stream'] = False / CAPTRA_L1_STREAM=0: off
        self.l1_stream = bool(cfg.get("l1_stream", True))
        # replay one captured hipGraph per frame instead of launching the ~140 kernels of a step one by one
        # (captra_amd/graph.py); opt-in: `--hipgraph` of captra_amd.track / cfg['hipgraph'].  Same kernels, same bits.
        self.use_graph = bool(cfg.get("hipgraph", False))
        # arithmetic of the shared MLPs for THIS model (None = whatever the calling thread has set, default exact fp32);
        # "bf16" = BASELINE.json configs[2].  Entered around every step (fused.use_mlp_dtype): no process-wide switch.
        self.mlp_dtype = cfg.get("mlp_dtype")
        # multi-GPU harness hooks (captra_amd/track.py): `frame_hook(i, pose)` is called with every frame's pose (B,P,...)
        # as soon as it is enqueued -- the per-frame all-gather of pose records starts there and runs under the next
        # frame's kernels; `result_sink(list of (file name, per-trajectory result dict))` receives what `_save` would
        # write (rank 0 writes the pickles of every rank's trajectories).  Both None = the single-process behaviour.
        self.frame_hook = None
        self.result_sink = None
        # nocs_otf at batch >= 32 as two lanes half a frame apart (_forward_otf_lanes): bit-identical results, 9.5 -> 7.9 ms per
        # 32-trajectory step (3370 -> 4060 frames/s).  On by default (cfg['otf_lanes'] = False turns it off): with the lane
        # streams created once per process the first eight model objects of a process all get the fast placement; what a
        # later one may get (both lanes on one hardware queue, ~13 ms) is in DESIGN.md section 5
        self.otf_lanes = bool(cfg.get("otf_lanes", True))
        self._graph = None
        self._graph_key = None

    # ---- host -> device ------------------------------------------------------------------------
    def _gt_part(self, frame):
        return part_model_batch_to_part(cvt_torch(frame["meta"]["nocs2camera"], self.device), self.num_parts, self.device)

    def _convert_pose_frame(self, frame, first):
        out = {"meta": frame["meta"], "gt_part": self._gt_part(frame)}
        if first:
            for key in ("points", "nocs"):
                if key in frame:
                    out[key] = frame[key].float().to(self.device)
        else:
            out["points"] = frame["points"].float().to(self.device)
            out["points_mean"] = frame["meta"]["points_mean"].float().to(self.device)
            if "nocs" in frame:
                out["npcs"] = frame["nocs"].float().to(self.device)
        if "labels" in frame:
            out["labels"] = frame["labels"].long().to(self.device)
        pre = frame["meta"].get("pre_fetched")
        if pre is not None:      # nocs_otf: the frame's depth image and instance mask live on the device
            out["pre_fetched"] = {"depth": torch.as_tensor(pre["depth"]).to(self.device).int(),
                                  "mask": torch.as_tensor(pre["mask"]).to(self.device).bool()}
            # the re-crop derives ground-truth NOCS from the root part's ground-truth pose: keep it on the host (float64,
            # the values of the float32 device copy), so that the loop does not fetch it back from the device every frame
            root = frame["meta"]["nocs2camera"][self.root]
            out["gt_root_host"] = {k: np.asarray(torch.as_tensor(root[k]).float().double().cpu().numpy()) for k in ("rotation", "translation", "scale")}
            out["gt_root_dev"] = {k: torch.from_numpy(v).to(self.device) for k, v in out["gt_root_host"].items()}   # (and on the device: no upload per frame)
        return out

    def _convert_npcs_frame(self, frame):
        out = {"meta": frame["meta"], "points_mean": frame["meta"]["points_mean"].float().to(self.device)}
        for key in ("points", "nocs"):
            if key in frame:
                out[key] = frame[key].float().to(self.device)
        if "labels" in frame:
            out["labels"] = frame["labels"].long().to(self.device)
        return out

    def set_data(self, data):
        self.feed_dict = [self._convert_pose_frame(f, i == 0) for i, f in enumerate(data)]
        self.npcs_feed_dict = [self._convert_npcs_frame(f) for f in data]

    # ---- the loop ------------------------------------------------------------------------------
    def _initial_pose(self):
        gt_part = self.feed_dict[0]["gt_part"]
        if self.gt_init:
            return gt_part
        part = add_noise_to_part_dof(gt_part, self.pose_perturb_cfg)
        if "crop_pose" in self.feed_dict[0]["meta"]:
            crop = part_model_batch_to_part(cvt_torch(self.feed_dict[0]["meta"]["crop_pose"], self.device),
                                            self.num_parts, self.device)
            part["translation"], part["scale"] = crop["translation"], crop["scale"]
        return part

    def track_step(self, input, npcs_input, last_pose):
        """One frame for all B trajectories: CoordNet -> labels -> RotationNet -> pose fit."""
        from . import fused
        with fused.use_mlp_dtype(self.mlp_dtype):
            return self._track_step(input, npcs_input, last_pose)

    def _track_step(self, input, npcs_input, last_pose):
        from . import fused
        few = (fused.split_k_rule(len(input["points"]), allow_few=not getattr(self, "_no_split_k", False))
               if not self.training and input["points"].is_cuda else 0)
        with fused.split_k(few):
            return self._track_step_body(input, npcs_input, last_pose)

    def _track_step_body(self, input, npcs_input, last_pose):
        self._step_begin(input, npcs_input, last_pose)
        join = self._fork_rotation_net(input, npcs_input, last_pose) if self._overlap_nets(input) else None
        if join is None and "_geom" not in npcs_input and self._l1_stream_on(npcs_input):
            # networks one after the other (no fork): the shared prefix with the level-1 stream kernel all the same -- CoordinateNet
            # takes it from `_geom`, RotationNet through the shared geometry
            self._step_prep(input, npcs_input, last_pose)
        npcs_pred = self._step_coord(npcs_input)
        if join is not None:
            join()
        return npcs_pred, self._step_post(input, npcs_input, npcs_pred, last_pose)

    # ---- the step in four phases (what `_track_step` composes; captra_amd.graph.TrackStepGraph(split=True) captures each as a
    # hipGraph of its own and replays [prep] -> [rot || coord] -> [post] on two EXPLICIT streams) ---------------------------
    def _step_begin(self, input, npcs_input, last_pose):
        # (a view when it is contiguous -- one part: the clones are three copy kernels per step and nothing writes into them)
        npcs_input["canon_pose"] = {k: (v if v.is_contiguous() else v.clone()) for k, v in
                                    ((k, last_pose[k][:, self.root]) for k in ("rotation", "translation", "scale"))}
        npcs_input["init_part"] = last_pose
        for k in ("_canon", "_geom"):
            npcs_input.pop(k, None)
        input.pop("_raw", None)

    def _step_prep(self, input, npcs_input, last_pose, level1_only=False, side=None) -> bool:
        """The part both networks wait for: CoordinateNet's canonicalised cloud and its geometry (sampling, neighbour lists,
        interpolation weights).  False when the cloud does not fit the one-launch sampler (no side-by-side schedule then)."""
        from . import fused
        from .networks import _canonicalize
        coord_bb = self.npcs_net.backbone
        # bf16 mode, one part: both networks' first level inside the sampler's launch (the level-1 stream kernel)
        stream = self._l1_stream_on(npcs_input)
        cam = _canonicalize(npcs_input["points"], npcs_input["points_mean"], npcs_input["canon_pose"], want_planes=stream)
        stream_level1 = None
        if stream:
            # RotationNet's backbone sees the bare coordinates, CoordinateNet's the coordinates as features too (use_xyz_feat)
            # RotationNet's backbone sees the bare coordinates, CoordinateNet's the coordinates as features too.  CAPTRA_L1_NETS=rot
            # (A/B): RotationNet's level only, CoordinateNet's three scales as launches of its own branch -- measured equal at 32
            # trajectories (1.27 ms per step either way: the step is bound by the chip's work, not by the sampler's latency)
            nets = [(self.net.regress_net.encoder, None)]
            if os.environ.get("CAPTRA_L1_NETS", "both") == "both":
                nets.append((coord_bb, cam[0]))
            stream_level1 = (cam[0], cam[2], nets)
        geom = coord_bb.precompute_geometry(cam[1], level1_only=level1_only, side=side, stream_level1=stream_level1)
        if geom is None:
            return False
        npcs_input["_canon"], npcs_input["_geom"] = (cam[0], cam[1]), geom
        scratch = (geom["sa1"].get("pooled") or {}).get("_scratch")
        if scratch is not None:
            # STICKY give-up word: the scratch is a fresh (eager) or re-zeroed (replayed) buffer every step, so its flag says
            # something about ONE launch; this one-word OR -- a launch of the step like any other, captured and replayed with it --
            # keeps every step's verdict until check_l1_stream() reads it (before results are written, and by the bench)
            self._l1_scratch = scratch
            sticky = getattr(self, "_l1_sticky", None)
            if (sticky is None or sticky.device != scratch.device) and not torch.cuda.is_current_stream_capturing():
                # (never created inside a capture: its zero fill would be replayed with the step; the loops run a step eagerly first)
                sticky = self._l1_sticky = torch.zeros(1, dtype=torch.int32, device=scratch.device)
            if sticky is not None and sticky.device == scratch.device:
                sticky.bitwise_or_(scratch.view(torch.int32)[-15:-14])
        return True

    def _l1_stream_on(self, npcs_input) -> bool:
        """bf16 mode, one part: both networks' first level inside the sampler's launch (the level-1 stream kernel)."""
        from . import fused
        pts = npcs_input["points"]
        return (self.l1_stream and self.num_parts == 1 and self.share_geometry and fused.USE_L1_STREAM and fused.mlp_dtype() == "bf16"
                and not self.training and pts.is_cuda and pts.shape[2] <= 4096 and pts.shape[0] <= fused.L1_STREAM_MAX_CLOUDS
                and not (self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]))

    def check_l1_stream(self) -> None:
        """Raises when a consumer of ANY level-1 stream launch since the last check gave up waiting for its sampler (bounded spins;
        synchronises): the sticky word every step ORs its flag into, then the last launch's own flag."""
        from . import fused
        sticky = getattr(self, "_l1_sticky", None)
        scratch = getattr(self, "_l1_scratch", None)
        bad = (sticky is not None and bool(sticky.item())) or (scratch is not None and fused.sa1_stream_gave_up(scratch))
        if sticky is not None:
            sticky.zero_()
        if bad:
            raise RuntimeError("level-1 stream kernel: a consumer workgroup gave up waiting for the sampler in a step of this run; "
                               "its outputs (and every pose after it) are incomplete")

    def _step_rot(self, input, npcs_input, last_pose):
        """RotationNet up to its heads' raw per-point output (needs nothing of CoordinateNet's but, for one part, its geometry)."""
        from .networks import _canonicalize
        P = self.num_parts
        cam, geom = npcs_input["_canon"], npcs_input["_geom"]
        if P == 1 and self.share_geometry:            # one part: RotationNet's cloud IS CoordinateNet's
            return self.net.regress_net.raw_point_rtvec(cam[0], cam_n3=cam[1], geom=geom)
        # every part's cloud canonicalised with that part's previous pose
        canon = {k: last_pose[k].reshape((-1,) + last_pose[k].shape[2:]) for k in ("rotation", "translation", "scale")}
        rcam = _canonicalize(input["points"], input["points_mean"], canon, num_parts=P)
        return self.net.regress_net.raw_point_rtvec(rcam[0], cam_n3=rcam[1])

    def _step_coord(self, npcs_input):
        return self.npcs_net(npcs_input)

    def _step_post(self, input, npcs_input, npcs_pred, last_pose):
        pred_npcs = npcs_pred["nocs"].reshape(len(npcs_pred["nocs"]), self.num_parts, 3, -1)
        input["state"] = {"part": last_pose}
        lab32 = npcs_pred.pop("_labels_i32", None)   # CoordinateNet's fused read-out: the int32 labels of THIS prediction
        if lab32 is not None and not (self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]):
            input["pred_labels_i32"] = lab32             # what the one-launch rotation read-out and pose fit take
            input["pred_labels"] = lab32 if self._overlap_nets(input) else lab32.long()
        else:
            input.pop("pred_labels_i32", None)
            input["pred_labels"] = torch.argmax(npcs_pred["seg"], dim=-2)
        input["pred_nocs"] = pred_npcs
        input["pred_label_conf"] = npcs_pred["seg"][:, 0]
        if self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]:
            input["pred_labels"] = npcs_input["labels"]
        input.pop("shared_geometry", None)
        if self.share_geometry and self.num_parts == 1 and not self.npcs_net.training:
            input["shared_geometry"] = (self.npcs_net.last_canon, self.npcs_net.backbone.last_geom)
        return self.net(input, test_mode=True)["part"]

    # ---- the two networks side by side -------------------------------------------------------------------------------
    def _overlap_nets(self, input) -> bool:
        """The two backbones do not depend on each other (RotationNet needs CoordNet's labels only for its read-out): they
        run on two streams = two branches of the captured graph, each filling the other's latency-bound stretches and
        launch ramps / tails.  1.37 -> 1.18 ms per frame at one trajectory, 6.56 -> 6.40 ms per step at 32."""
        from . import fused
        return (self.overlap_nets
                and not self.training and input["points"].is_cuda and fused.USE_ROT_READOUT
                and not (self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]) and not self.net.return_point_rotation)

    def _fork_rotation_net(self, input, npcs_input, last_pose):
        small = len(input["points"]) <= 2      # one or two trajectories: every kernel is latency-bound, overlap all that can be (no gain from 4 up)
        dev = input["points"].device
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side
        main = torch.cuda.current_stream(dev)
        gstream = None
        if self.sampler_chunks > 1 and self.num_parts == 1 and self.share_geometry:
            # the geometry on a stream of its own, both networks forked right behind the canonicalisation: their first level
            # waits for the sampler's parts one by one, their second level for the rest of the geometry
            from .networks import _canonicalize
            if getattr(self, "_gstream", None) is None:
                self._gstream = torch.cuda.Stream(device=dev)
            cam = _canonicalize(npcs_input["points"], npcs_input["points_mean"], npcs_input["canon_pose"])
            geom = self.npcs_net.backbone.precompute_geometry_streamed(cam[1], self.sampler_chunks, self._gstream, consumers=(side,),
                                                                       backbones=(self.net.regress_net.encoder,))
            if geom is not None:
                npcs_input["_canon"], npcs_input["_geom"] = cam, geom
                gstream = self._gstream
        if gstream is None and not self._step_prep(input, npcs_input, last_pose, level1_only=small, side=side if self.overlap_geometry else None):
            return None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            raw = self._step_rot(input, npcs_input, last_pose)

        def join():
            main.wait_stream(side)
            if gstream is not None:
                main.wait_stream(gstream)
            raw.record_stream(main)
            input["_raw"] = raw
        return join

    def _graph_usable(self, input) -> bool:
        return (self.use_graph and not self.training and input["points"].is_cuda
                and not (self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]))

    def _graph_step(self, input, last_pose):
        """One frame through the captured graph (captured on first use for this batch shape); outputs are cloned out
        of the graph's static buffers."""
        from .graph import TrackStepGraph
        key = (tuple(input["points"].shape), str(input["points"].device))
        if self._graph is None or self._graph_key != key or self._graph.stale():
            self._graph = TrackStepGraph(self, input["points"], input["points_mean"], last_pose)
            self._graph_key = key
        pose = self._graph.replay(input["points"], input["points_mean"], last_pose)
        npcs = {k: v.clone() for k, v in self._graph.npcs_pred.items() if torch.is_tensor(v)}
        return npcs, {k: v.clone() for k, v in pose.items()}

    def _lanes_usable(self, input) -> bool:
        """From 32 trajectories on the captured step runs as two free-running lanes (graph.TrackLanes; +2.8 % frames/s,
        -1 % at 8 and 16).  Not with the on-the-fly re-crop: it reads the previous pose on the host every frame."""
        B = input["points"].shape[0]
        return self._graph_usable(input) and not self.nocs_otf and B >= 32 and B % 2 == 0

    def _lanes_for(self, input, pose):
        from .graph import TrackLanes
        key = ("lanes", tuple(input["points"].shape), str(input["points"].device))
        if self._graph is None or self._graph_key != key or self._graph.stale():
            self._graph = TrackLanes(self, input["points"], input["points_mean"], pose, lanes=2, keep_npcs=True)
            self._graph_key = key
        else:
            self._graph.set_pose(pose)
        return self._graph

    def _recrop_slice(self, i, input, last_pose, sl, defer=None):
        """nocs_otf (reference model.py:425-452) for the trajectories `sl` of frame i: re-crop around the pose predicted for
        frame i-1 -- on the device (captra_amd/nocs_otf.py: one crop launch + one ragged sampling launch).  `last_pose` holds
        those trajectories only.  -> (points (b,3,N) mean-subtracted, labels (b,N), nocs (b,3,N)).  With the pose on the device the
        stage's one host round trip is the crops' member counts; `defer` (an int: upper bound of the candidate lists' length) takes
        that one away too (nocs_otf.full_data_batch_arrays) and appends the device word [rare-path instance met, longest list] to
        the result -- the caller reads it a frame late (_OtfCheck) and runs the frame again without `defer` when it is set."""
        from .nocs_otf import full_data_batch_arrays, to_host
        pre = input.get("pre_fetched")
        if pre is None:
            raise ValueError("nocs_otf=True needs the frame's depth and mask tensors (meta['pre_fetched']): reading depth.png / "
                             "mask.png from disk (cv2) is outside this build")
        npcs = self.npcs_feed_dict[i]
        N = input["points"].shape[2]
        b = last_pose["scale"].shape[0]
        gt = input.get("gt_root_host")
        if gt is None:
            gt = {k: to_host(v[:, self.root].double().contiguous()) for k, v in input["gt_part"].items()}
        gt = {k: v[sl] for k, v in gt.items()}
        depth, mask = pre["depth"][sl], pre["mask"][sl]
        gt64 = {"rotation": np.asarray(gt["rotation"], np.float64).reshape(b, 3, 3),
                "translation": np.asarray(gt["translation"], np.float64).reshape(b, 3),
                "scale": np.asarray(gt["scale"], np.float64).reshape(b)}
        gtd = input.get("gt_root_dev")
        gtd = None if gtd is None else {k: v[sl] for k, v in gtd.items()}
        trans_d, scale_d = last_pose["translation"][:, self.root].reshape(b, 3), last_pose["scale"][:, self.root].reshape(b)
        if OTF_POSE_ON_DEVICE and depth.is_cuda and trans_d.dtype == torch.float32 and scale_d.dtype == torch.float32:
            # the crop's box / centre / radius derived on the device from the pose (captra_crop_box): no round trip for the pose
            full = full_data_batch_arrays(depth, mask, None, None, gt64, N, stacked=True, pose_dev=(trans_d, scale_d, float(self.radius)), gt_dev=gtd,
                                          defer=defer, mean=npcs["points_mean"][sl])
            if defer is not None:
                return full["points_cn"], full["labels"], full["nocs_cn"], full["_info"]
        else:
            defer = None
            cs = to_host(torch.cat([trans_d, scale_d.reshape(b, 1)], dim=1).double())
            full = full_data_batch_arrays(depth, mask, cs[:, :3], self.radius * cs[:, 3], gt64, N, stacked=True)
        points = (full["points"].float() - npcs["points_mean"][sl].reshape(b, 1, 3)).transpose(1, 2).contiguous()
        return points, full["labels"].contiguous(), full["nocs"].float().transpose(1, 2).contiguous()

    def _recrop(self, i, input, last_pose, defer=None):
        """The whole batch of frame i re-cropped in place (input / npcs feed dicts); -> the deferred check's device word or None."""
        npcs = self.npcs_feed_dict[i]
        res = self._recrop_slice(i, input, last_pose, slice(None), defer=defer)
        input["points"], input["labels"], npcs["nocs"] = res[:3]
        npcs["points"], npcs["labels"] = input["points"], input["labels"]
        return res[3] if len(res) > 3 else None

    def _otf_defer_usable(self, input) -> bool:
        """The re-crop without its round trip: pose on the device, no per-frame hook that publishes a frame's pose at once (the
        distributed harness's exchange: a frame that is run again a frame later would already be out)."""
        return (OTF_DEFER and OTF_POSE_ON_DEVICE and self.nocs_otf and input["points"].is_cuda and not self.training
                and self.frame_hook is None)

    # ---- nocs_otf at batch >= 32: two lanes of trajectories, half a frame apart ------------------------------------------
    def _otf_lanes_usable(self, input) -> bool:
        """The re-crop's sampler (<= 20480 -> 4096 points: 4095 dependent rounds, ONE workgroup per trajectory) leaves
        240 of the 256 CUs idle for 2.8 ms of a 10.8 ms step.  With the batch split into two lanes on two streams, started
        half a cycle apart, one lane samples while the other runs its networks.  The host stays single-threaded: it serves
        lane 0's frame i+1 as soon as lane 0's pose i is back (lane 1's frame i is executing meanwhile), then lane 1's."""
        B = input["points"].shape[0]
        return (self.nocs_otf and input["points"].is_cuda and not self.training and B >= 32 and B % 2 == 0
                and not (self.track_cfg["gt_label"] or self.track_cfg["nocs2d_label"]))

    def _forward_otf_lanes(self, pose0, frame_nums):
        from .graph import TrackStepGraph
        feed = self.feed_dict
        B = feed[1]["points"].shape[0]
        half = B // 2
        slices = [slice(0, half), slice(half, B)]
        dev = feed[1]["points"].device
        cur = torch.cuda.current_stream(dev)
        from . import graph as G
        if G.SPLIT_OTF_LANES:
            pairs = G.lane_streams(dev, 2)       # process-wide explicit streams: see captra_amd/graph.py SPLIT_OTF_LANES
            streams, sides = [m_ for m_, _ in pairs], [s_ for _, s_ in pairs]
        else:
            if getattr(self, "_otf_streams", None) is None:
                key = (dev.index if dev.index is not None else torch.cuda.current_device())
                if key not in _OTF_LANE_STREAMS:
                    _OTF_LANE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in slices]
                self._otf_streams = _OTF_LANE_STREAMS[key]
            streams, sides = self._otf_streams, [None, None]
        use_graph = self._graph_usable(feed[1])
        graphs = None
        if use_graph:
            key = ("otf", tuple(feed[1]["points"].shape), str(dev))
            if self._graph is None or self._graph_key != key or any(g.stale() for g in self._graph):
                self._graph = [TrackStepGraph(self, feed[1]["points"][s].contiguous(), feed[1]["points_mean"][s].contiguous(),
                                              {k: v[s].contiguous() for k, v in pose0.items()}, split_side=side, allow_split_k=False) for s, side in zip(slices, sides)]
                self._graph_key = key
            graphs = self._graph
        lane_pose = [{k: v[s].clone() for k, v in pose0.items()} for s in slices]
        for st in streams:
            st.wait_stream(cur)
        pred_poses, npcs_pred = [pose0], [None]
        state = {"sampled": None}      # recorded on a lane's stream when its re-crop (crop + sampling launch) of the current frame is enqueued
        N = feed[1]["points"].shape[2]
        defer_ok = self._otf_defer_usable(feed[1])

        def run_frame(i, poses_in, bounds):
            """Frame i of both lanes from the poses entering it; bounds[l] = the sync-free re-crop's stride bound of lane l or None
            (the synchronous stage).  -> (parts, events, poses out, deferred checks)."""
            input = feed[i]
            parts, done, poses_out, checks = [], [], [], []
            for l, s in enumerate(slices):
                with torch.cuda.stream(streams[l]):
                    if state["sampled"] is not None:
                        # the other lane's sampling of ITS current frame is over before this lane starts to sample: the two
                        # samplers never share the chip (each runs under the other lane's networks), whatever phase the lanes
                        # would drift into by themselves.  A GPU-side wait: the host goes on enqueuing.
                        if bounds[l] is not None:
                            streams[l].wait_event(state["sampled"])
                        else:
                            state["sampled"].synchronize()
                    res = self._recrop_slice(i, input, poses_in[l], s, defer=bounds[l])
                    pts, labels, nocs = res[:3]
                    checks.append(_OtfCheck(res[3]) if len(res) > 3 and res[3] is not None else None)
                    state["sampled"] = torch.cuda.Event()
                    state["sampled"].record(streams[l])
                    mean = input["points_mean"][s]
                    if graphs is not None:
                        out = graphs[l].replay(pts, mean, poses_in[l])
                        pose = {k: v.clone() for k, v in out.items()}
                        cur_npcs = {k: v.clone() for k, v in graphs[l].npcs_pred.items() if torch.is_tensor(v)}
                    else:
                        lin = {"points": pts, "points_mean": mean, "meta": {}, "labels": labels}
                        lnp = {"points": pts, "points_mean": mean, "labels": labels}
                        cur_npcs, pose = self.track_step(lin, lnp, poses_in[l])
                        cur_npcs = {k: v for k, v in cur_npcs.items() if torch.is_tensor(v)}
                    poses_out.append(pose)
                    ev = torch.cuda.Event()
                    ev.record(streams[l])
                parts.append((pts, labels, nocs, pose, cur_npcs))
                done.append(ev)
            return parts, done, poses_out, checks

        def commit(i, parts, done, replace):
            """The frame's batch-wide tensors, assembled on the caller's stream (GPU-side waits: the lanes do not stop)."""
            input, npcs_in = feed[i], self.npcs_feed_dict[i]
            for ev in done:
                cur.wait_event(ev)
            for part in parts:
                for x in part[:3]:
                    x.record_stream(cur)
                for d in part[3:]:
                    for v in d.values():
                        v.record_stream(cur)
            input["points"] = torch.cat([p[0] for p in parts])
            input["labels"] = torch.cat([p[1] for p in parts])
            npcs_in["nocs"] = torch.cat([p[2] for p in parts])
            npcs_in["points"], npcs_in["labels"] = input["points"], input["labels"]
            pose = {k: torch.cat([p[3][k] for p in parts]) for k in parts[0][3]}
            npcs = {k: torch.cat([p[4][k] for p in parts]) for k in parts[0][4]}
            if replace:
                npcs_pred[i], pred_poses[i] = npcs, pose
            else:
                npcs_pred.append(npcs)
                pred_poses.append(pose)
            return pose

        bounds = [OTF_FIRST_BOUND * N if defer_ok else None] * len(slices)
        pending = None                  # (frame, poses that entered it, its deferred checks)
        for i in range(1, len(feed)):
            frame_nums.append([p.split(".")[-2].split("/")[-1] for p in feed[i]["meta"]["path"]])
            if pending is not None:
                pi, pin, checks = pending
                pending = None
                verdicts = [c.read() for c in checks if c is not None]
                if any(r for r, _ in verdicts):
                    # a rare-path instance in the previous frame: that frame once more, both lanes, on the synchronous stage
                    parts, done, lane_pose, _ = run_frame(pi, pin, [None] * len(slices))
                    commit(pi, parts, done, replace=True)
                elif verdicts:
                    bounds = [_otf_bound(longest, N) for _, longest in verdicts]
            consume_noise_draws(feed[i - 1]["gt_part"], self.pose_perturb_cfg)      # (after a replay: see forward())
            poses_in = lane_pose
            parts, done, lane_pose, checks = run_frame(i, poses_in, bounds)
            pose = commit(i, parts, done, replace=False)
            if any(c is not None for c in checks):
                pending = (i, poses_in, checks)
            if self.frame_hook is not None:
                self.frame_hook(i, pose)
        if pending is not None:
            pi, pin, checks = pending
            if any(c.read()[0] for c in checks if c is not None):
                parts, done, lane_pose, _ = run_frame(pi, pin, [None] * len(slices))
                commit(pi, parts, done, replace=True)
        for st in streams:
            cur.wait_stream(st)
        return pred_poses, npcs_pred

    def forward(self, save=False):
        pred_poses = [self._initial_pose()]
        if self.frame_hook is not None:
            self.frame_hook(0, pred_poses[0])
        npcs_pred = [None]
        frame_nums = []
        self.timer.tick()
        lanes = None
        if len(self.feed_dict) > 1 and self._otf_lanes_usable(self.feed_dict[1]) and self.otf_lanes:
            frame_nums.append([p.split(".")[-2].split("/")[-1] for p in self.feed_dict[0]["meta"]["path"]])
            with torch.no_grad():
                pred_poses, npcs_pred = self._forward_otf_lanes(pred_poses[0], frame_nums)
            self.pred_dict = {"poses": pred_poses, "npcs_pred": npcs_pred}
            if save:
                self._save(frame_nums)
            return
        with torch.no_grad():
            if len(self.feed_dict) > 1 and self._lanes_usable(self.feed_dict[1]):
                lanes = self._lanes_for(self.feed_dict[1], pred_poses[0])
            pending, bound = None, OTF_FIRST_BOUND * self.feed_dict[0]["points"].shape[2]
            for i, input in enumerate(self.feed_dict):
                frame_nums.append([p.split(".")[-2].split("/")[-1] for p in input["meta"]["path"]])
                if i == 0:
                    continue
                if lanes is not None:
                    # the lanes hand their poses over themselves; this stream only copies the frame's records out
                    consume_noise_draws(self.feed_dict[i - 1]["gt_part"], self.pose_perturb_cfg)
                    pose, cur_npcs = lanes.gather(lanes.step(input["points"], input["points_mean"], sync_inputs=(i == 1)), npcs=True)
                    npcs_pred.append({k: v.clone() for k, v in cur_npcs.items()})
                    pred_poses.append({k: v.clone() for k, v in pose.items()})
                    if self.frame_hook is not None:
                        self.frame_hook(i, pred_poses[-1])
                    continue
                def run(fi, finput, defer):
                    lp = {k: v.clone() for k, v in pred_poses[fi - 1].items()}
                    info = self._recrop(fi, finput, lp, defer=defer) if self.nocs_otf else None
                    if self._graph_usable(finput):
                        out = self._graph_step(finput, lp)
                    else:
                        out = self.track_step(finput, self.npcs_feed_dict[fi], lp)
                    return out, info

                if pending is not None:
                    # the PREVIOUS frame's deferred verdict: a rare-path instance -> that frame once more, synchronously
                    rare, longest = pending.read()
                    pending = None
                    if rare:
                        (npcs_pred[i - 1], pred_poses[i - 1]), _ = run(i - 1, self.feed_dict[i - 1], None)
                    else:
                        bound = _otf_bound(longest, input["points"].shape[2])
                # the reference draws (and discards) a perturbed pose every frame (model.py:414); draw it too so that seeded runs
                # consume the generator identically -- AFTER a replay of the previous frame (its thinning permutations come out of
                # the same numpy generator and precede this frame's draw in the reference's order)
                consume_noise_draws(self.feed_dict[i - 1]["gt_part"], self.pose_perturb_cfg)
                defer = bound if (self.nocs_otf and self._otf_defer_usable(input)) else None
                (cur_npcs, pose), info = run(i, input, defer)
                if info is not None:
                    pending = _OtfCheck(info)
                npcs_pred.append(cur_npcs)
                pred_poses.append(pose)
                if self.frame_hook is not None:
                    self.frame_hook(i, pose)
            if pending is not None and pending.read()[0]:
                last = len(self.feed_dict) - 1
                (npcs_pred[last], pred_poses[last]), _ = run(last, self.feed_dict[last], None)
        self.pred_dict = {"poses": pred_poses, "npcs_pred": npcs_pred}
        self.check_l1_stream()
        if save:
            self._save(frame_nums)

    def _save(self, frame_nums):
        """Per-trajectory pickle {'pred': {'poses','corners'}, 'gt': {'poses','corners'}, 'frame_nums'}
        (reference model.py:482-509).  Predicted NOCS corners = `get_pred_nocs_corners` of the points' own-part predicted
        coordinates: per part the symmetric extent [-max|x|, +max|x|] (model.py:489-493) -- the same boxes compute_loss
        evaluates, so the offline IoU tables (captra_amd/eval.py) agree with the in-loop avg_iou."""
        from .loss import choose_coord_by_label
        from .pose_utils.bbox_utils import get_pred_nocs_corners
        gt_corners = self.feed_dict[0]["meta"]["nocs_corners"].cpu().numpy()
        corner_list = [None]
        for i in range(1, len(self.pred_dict["poses"])):
            pred = self.pred_dict["npcs_pred"][i]
            pred_labels = torch.max(pred["seg"], dim=-2)[1]                                        # (B,N)
            pred_nocs = choose_coord_by_label(pred["nocs"].transpose(-1, -2), pred_labels)         # (B,N,3)
            corner_list.append(get_pred_nocs_corners(pred_labels, pred_nocs, self.num_parts))
        to_np = lambda pose: {k: v.detach().cpu().numpy() for k, v in pose.items()}
        save_dict = {"pred": {"poses": [to_np(p) for p in self.pred_dict["poses"]], "corners": corner_list},
                     "gt": {"poses": [to_np(f["gt_part"]) for f in self.feed_dict], "corners": gt_corners},
                     "frame_nums": frame_nums}
        records = []
        for i, path in enumerate(self.feed_dict[0]["meta"]["path"]):
            instance, track_num = path.split(".")[-2].split("/")[-3:-1]
            records.append((f"{instance}_{track_num}.pkl", get_ith_from_batch(save_dict, i, to_single=False)))
        if self.result_sink is not None:
            self.result_sink(records)
        else:
            write_result_pickles(self.cfg["experiment_dir"], records)

    def compute_loss(self, test=False, per_instance=False, eval_iou=False, test_prefix=None):
        """The reference's compute_loss (model.py:511-593): per-part rdiff / tdiff / sdiff / 5deg5cm averaged over frames
        1..T-1 for the prediction and for its initialisation (= the previous frame's prediction), the segmentation and
        NOCS losses of CoordinateNet's maps when the frames carry labels / NOCS, and with `eval_iou` the three box IoUs
        (canonical boxes, posed predicted box, ground-truth box under the predicted pose; host-side numpy as in the
        reference).  Keys as in the reference, including its quirk of storing the per-frame NOCS losses under
        'frame_seg'."""
        from .loss import choose_coord_by_label, compute_miou_loss, compute_nocs_loss
        from .pose_utils.bbox_utils import eval_single_part_iou, get_pred_nocs_corners
        avg_pred, avg_init, all_pred, all_init = {}, {}, {}, {}
        avg_iou, all_iou, seg_losses, all_seg, nocs_losses, all_nocs = {}, {}, [], {}, [], {}
        poses = self.pred_dict["poses"]
        gt_corners = self.feed_dict[0]["meta"]["nocs_corners"].float().cpu()                       # (B,P,2,3)
        for i, pose in enumerate(poses):
            diff, per = eval_part_full(self.feed_dict[i]["gt_part"], pose, per_instance=per_instance, yaxis_only=self.sym)
            all_pred[i] = deepcopy(diff)
            if i == 0:
                continue
            add_dict(avg_pred, diff)
            if per_instance:
                self.record_per_diff(self.feed_dict[i], per)
            init_diff, _ = eval_part_full(self.feed_dict[i]["gt_part"], poses[i - 1], per_instance=False, yaxis_only=self.sym)
            add_dict(avg_init, init_diff)
            all_init[i] = deepcopy(init_diff)
            npcs_pred, npcs_feed = self.pred_dict["npcs_pred"][i], self.npcs_feed_dict[i]
            if npcs_pred is None:
                continue
            if "labels" in npcs_feed:
                all_seg[i] = compute_miou_loss(npcs_pred["seg"], npcs_feed["labels"].long(), per_instance=False)
                seg_losses.append(all_seg[i])
            pred_labels = torch.max(npcs_pred["seg"], dim=-2)[1]
            if "nocs" in npcs_feed:
                all_nocs[i] = compute_nocs_loss(npcs_pred["nocs"], npcs_feed["nocs"], labels=pred_labels, confidence=None,
                                                loss="l2", self_supervise=False, per_instance=False)
                nocs_losses.append(all_nocs[i])
            if eval_iou:
                pred_nocs = choose_coord_by_label(npcs_pred["nocs"].transpose(-1, -2), pred_labels)                 # (B,N,3)
                pred_corners = torch.from_numpy(get_pred_nocs_corners(pred_labels, pred_nocs, self.num_parts)).float()
                iou, per_iou = eval_single_part_iou(gt_corners, pred_corners, self.feed_dict[i]["gt_part"], pose,
                                                    separate="both", nocs=self.nocs_otf, sym=self.sym)
                iou = {name: {p: float(v) for p, v in d.items()} for name, d in iou.items()}
                add_dict(avg_iou, iou)
                if per_instance:
                    self.record_per_diff(self.feed_dict[i], per_iou)
                all_iou[i] = deepcopy(iou)
        n = max(len(poses) - 1, 1)
        loss_dict = {"avg_pred": divide_dict(avg_pred, n), "avg_init": divide_dict(avg_init, n),
                     "frame_pred": all_pred, "frame_init": all_init}
        if seg_losses:
            loss_dict.update({"avg_seg": torch.mean(torch.stack(seg_losses)), "frame_seg": all_seg})
        if nocs_losses:
            loss_dict.update({"avg_nocs": torch.mean(torch.stack(nocs_losses)), "frame_seg": all_nocs})
        if eval_iou:
            loss_dict.update({"avg_iou": divide_dict(avg_iou, n), "frame_iou": all_iou})
        self.loss_dict = loss_dict

    def test(self, save=False, no_eval=False, epoch=0):
        self.forward(save=save)
        if no_eval:
            self.loss_dict = {}
        else:
            self.compute_loss(test=True, per_instance=save, eval_iou=True, test_prefix="test")
