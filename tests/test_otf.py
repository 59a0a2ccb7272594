"""On-the-fly ball crop + resample (captra_amd/nocs_otf.py) against golden G11, produced by the reference's own
crop_ball_from_depth_image + base_generate_data (tests/golden/make_golden_otf.py).  The CPU test injects the oracle FPS;
the GPU test runs the product path (device tensors, captra_fps_gather)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from captra_amd import nocs_otf
from tests.golden.make_golden_otf import CASES, make_frame

G = np.load(Path(__file__).resolve().parent / "golden" / "g11_otf.npz")


def _oracle_fps(points_f32: torch.Tensor, num: int) -> torch.Tensor:
    from oracle import ops as O
    return torch.from_numpy(O.furthest_point_sample(points_f32.cpu().numpy()[None], num)[0].astype(np.int64)).to(points_f32.device)


def _check(tag, seed, radius, n, device, fps_fn):
    depth, mask, center, pose = make_frame(seed)
    np.testing.assert_array_equal(nocs_otf.proj_corners(depth.shape[0], depth.shape[1], center, radius), G[f"{tag}_corners"])
    np.random.seed(100 + seed)
    kw = {} if fps_fn is None else {"fps_fn": fps_fn}
    full = nocs_otf.full_data_from_depth(torch.from_numpy(depth.astype(np.int32)).to(device), torch.from_numpy(mask).to(device),
                                         center, radius, pose, n, **kw)
    assert full["points"].shape == (n, 3) and full["points"].dtype == torch.float64
    if str(device) == "cpu":
        np.testing.assert_array_equal(full["points"].cpu().numpy(), G[f"{tag}_points"])  # same pixels, same float64 arithmetic
    else:   # the device's float64 3x3 product may contract into FMAs: last-bit differences, same pixels selected
        np.testing.assert_allclose(full["points"].cpu().numpy(), G[f"{tag}_points"], atol=1e-15, rtol=0)
    np.testing.assert_array_equal(full["labels"].cpu().numpy(), G[f"{tag}_labels"])
    np.testing.assert_allclose(full["nocs"].cpu().numpy(), G[f"{tag}_nocs"], atol=1e-12, rtol=0)


@pytest.mark.parametrize("tag,seed,radius,n", CASES)
def test_crop_and_resample_vs_reference_cpu(tag, seed, radius, n):
    _check(tag, seed, radius, n, "cpu", _oracle_fps)


def test_proj_corners_batch_equals_per_instance_cpu():
    """The track loop's vectorised box projection (nocs_otf.proj_corners_batch, written without a matrix product) == the
    per-instance form pinned by G11, on the fixture frames and on random centres / radii incl. boxes clamped at the image
    border and radii below the 0.05 floor."""
    rng = np.random.default_rng(5)
    centers = [make_frame(seed)[2] for _, seed, _, _ in CASES]
    radii = [r for _, _, r, _ in CASES]
    for _ in range(200):
        centers.append(np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.45, 0.45), -rng.uniform(0.4, 2.5)]))
        radii.append(float(rng.choice([0.01, 0.05, 0.12, 0.3, 0.8])))
    got = nocs_otf.proj_corners_batch(480, 640, np.stack(centers), np.asarray(radii))
    for i, (c, r) in enumerate(zip(centers, radii)):
        np.testing.assert_array_equal(got[i], nocs_otf.proj_corners(480, 640, c, r), err_msg=str((c, r)))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,seed,radius,n", CASES)
def test_crop_and_resample_vs_reference_gpu(device, tag, seed, radius, n):
    _check(tag, seed, radius, n, device, None)


@pytest.mark.gpu
def test_track_loop_with_on_the_fly_crop(device):
    """EvalTrackModel with nocs_otf=True: every frame's cloud is re-cropped on the device around the previous pose; with
    init_frame/gt the first re-crop is centred on the ground-truth pose of frame 0, so it equals a direct call."""
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    from tests import clouds
    from tests.weights import make_state_dict
    cfg = make_config("1", experiment_dir="/tmp/captra_otf_test", nocs_otf=True, **{"init_frame/gt": True})
    cfg["device"] = device
    trainer = Trainer(cfg)
    sd = make_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, seed=7)
    trainer.model.load_state_dict(sd)
    frames = clouds.make_trajectory("nocs", 2, 3, seed=0)
    depth, mask, center, pose = make_frame(1)
    for f in frames:                                        # same synthetic depth frame for every trajectory and time step
        f["meta"]["pre_fetched"] = {"depth": torch.from_numpy(np.stack([depth.astype(np.int32)] * 2)), "mask": torch.from_numpy(np.stack([mask] * 2))}
        for p in f["meta"]["nocs2camera"]:
            p["rotation"] = torch.from_numpy(np.stack([pose["rotation"]] * 2)).float()
            p["translation"] = torch.from_numpy(np.stack([pose["translation"]] * 2)).float()
            p["scale"] = torch.full((2,), float(pose["scale"]))
    np.random.seed(5)
    pred, _ = trainer.test(frames)
    assert len(pred["poses"]) == 3 and all(torch.isfinite(v).all() for p in pred["poses"] for v in p.values())
    model = trainer.model
    np.random.seed(5)
    c0 = pose["translation"].reshape(3).astype(np.float32).astype(np.float64)
    ref = nocs_otf.full_data_from_depth(torch.from_numpy(depth.astype(np.int32)).to(device), torch.from_numpy(mask).to(device), c0,
                                        cfg["data_radius"] * float(np.float32(pose["scale"])), pose, 4096)
    got = model.feed_dict[1]["points"][0].t().double() + model.npcs_feed_dict[1]["points_mean"][0].reshape(1, 3).double()
    np.testing.assert_allclose(got.cpu().numpy(), ref["points"].cpu().numpy(), atol=2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("hipgraph", [False, True])
def test_track_loop_otf_lanes_equal_single_batch(device, hipgraph):
    """nocs_otf at 32 trajectories: the two lanes half a frame apart (EvalTrackModel._forward_otf_lanes: one lane re-crops and
    samples while the other runs its networks) give the SAME poses, CoordinateNet maps and re-cropped clouds as the whole
    batch processed in one piece -- bit for bit, eager and with captured steps; the loss dict (IoUs included) agrees."""
    from captra_amd.configs import make_config
    from captra_amd.trainer import Trainer
    from tests import clouds
    from tests.weights import make_physical_state_dict
    B, T = 32, 4
    cfg = make_config("1", experiment_dir="/tmp/captra_otf_lanes_test", nocs_otf=True, hipgraph=hipgraph, **{"init_frame/gt": True})
    cfg["device"] = device
    trainer = Trainer(cfg)
    trainer.model.load_state_dict(make_physical_state_dict({k: tuple(v.shape) for k, v in trainer.model.state_dict().items()}, 7, 1, True, "nocs"))
    trainer.model.use_graph = hipgraph
    frames = clouds.make_trajectory("nocs", B, T, seed=2)
    views = [make_frame(1 + b % 3) for b in range(B)]            # three different synthetic depth frames over the batch
    for f in frames:
        f["meta"]["pre_fetched"] = {"depth": torch.from_numpy(np.stack([v[0].astype(np.int32) for v in views])),
                                    "mask": torch.from_numpy(np.stack([v[1] for v in views]))}
        for p in f["meta"]["nocs2camera"]:
            p["rotation"] = torch.from_numpy(np.stack([v[3]["rotation"] for v in views])).float()
            p["translation"] = torch.from_numpy(np.stack([v[3]["translation"] for v in views])).float()
            p["scale"] = torch.tensor([float(v[3]["scale"]) for v in views])
    runs = []
    for lanes in (False, True):
        trainer.model.otf_lanes = lanes
        np.random.seed(5)
        torch.manual_seed(5)
        pred, loss = trainer.test(frames, save=False, no_eval=False)
        runs.append(([{k: v.cpu().numpy() for k, v in p.items()} for p in pred["poses"]],
                     [None if n is None else {k: v.cpu().numpy() for k, v in n.items()} for n in pred["npcs_pred"]],
                     [trainer.model.feed_dict[i]["points"].cpu().numpy() for i in range(1, T)], loss))
    (p0, n0, c0, l0), (p1, n1, c1, l1) = runs
    for i in range(T):
        for k in p0[i]:
            np.testing.assert_array_equal(p0[i][k], p1[i][k], err_msg=f"frame {i} {k}")
        if n0[i] is not None:
            for k in n1[i]:
                np.testing.assert_array_equal(n0[i][k], n1[i][k], err_msg=f"frame {i} npcs {k}")
    for a, b in zip(c0, c1):
        np.testing.assert_array_equal(a, b)
    for k, v in l0["avg_pred"].items():
        assert abs(float(v) - float(l1["avg_pred"][k])) < 1e-6
    assert min(float(p["scale"].min()) for p in p1) > 0.05


def _items(device, cases):
    items = []
    for tag, seed, radius, n in cases:
        depth, mask, center, pose = make_frame(seed)
        items.append((torch.from_numpy(depth.astype(np.int32)).to(device), torch.from_numpy(mask).to(device), center, radius, pose))
    return items


def _same(a, b):
    np.testing.assert_allclose(a["points"].cpu().numpy(), b["points"].cpu().numpy(), atol=1e-15, rtol=0)
    np.testing.assert_array_equal(a["labels"].cpu().numpy(), b["labels"].cpu().numpy())
    np.testing.assert_allclose(a["nocs"].cpu().numpy(), b["nocs"].cpu().numpy(), atol=1e-12, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("use_kernel", [False, True])
def test_batched_recrop_equals_one_call_per_trajectory(device, use_kernel):
    """full_data_batch (one crop launch + one ragged furthest-point-sampling launch for all trajectories of a step) ==
    full_data_from_depth called once per trajectory, including the order in which the thinning permutations are drawn.
    use_kernel=False keeps the candidate extraction in torch ops (bit-identical); the crop kernel writes the float64
    back-projection out operation by operation (last-bit differences against torch's matmul, same pixels)."""
    n = CASES[0][3]
    items = _items(device, [c for c in CASES if c[3] == n])
    items = items + [items[0]]
    np.random.seed(11)
    one_by_one = [nocs_otf.full_data_from_depth(d, m, c, r, p, n) for d, m, c, r, p in items]
    np.random.seed(11)
    batched = nocs_otf.full_data_batch(items, n, use_kernel=use_kernel)
    for a, b in zip(one_by_one, batched):
        if use_kernel:
            _same(a, b)
        else:
            for k in ("points", "labels", "nocs"):
                assert torch.equal(a[k], b[k])


@pytest.mark.gpu
def test_crop_box_on_device_equals_host_projection(device):
    """captra_crop_box (box, centre, radius of the crop from the device-resident fp32 pose) == nocs_otf.proj_corners_batch and the
    float64 casts the loop did on the host, bit for bit, on poses around the camera frustum (boxes clamped on every side, radius
    clamped to 0.05) -- and the re-crop through it equals the re-crop through the host, including an instance on the rare path."""
    from captra_amd import _lib as L
    rng = np.random.default_rng(5)
    B, H, W = 257, 480, 640
    trans = np.stack([rng.uniform(-0.6, 0.6, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-2.5, -0.4, B)], 1).astype(np.float32)
    scale = rng.uniform(0.01, 0.6, B).astype(np.float32)
    factor = 0.6
    t_d, s_d = torch.from_numpy(trans).to(device), torch.from_numpy(scale).to(device)
    kk = nocs_otf._intrinsics_on_device(nocs_otf.NOCS_REAL_INTRINSICS, torch.device(device))
    box = torch.empty(B, 4, dtype=torch.int32, device=device)
    ctr = torch.empty(B, 3, dtype=torch.float64, device=device)
    rad = torch.empty(B, dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        L.call("captra_crop_box", B, H, W, factor, L.ptr(t_d), L.ptr(s_d), kk.data_ptr(), L.ptr(box), L.ptr(ctr), L.ptr(rad))
    c64, r_in = trans.astype(np.float64), factor * scale.astype(np.float64)
    want = nocs_otf.proj_corners_batch(H, W, c64, r_in).reshape(B, 4)
    np.testing.assert_array_equal(box.cpu().numpy(), want)
    np.testing.assert_array_equal(ctr.cpu().numpy(), c64)
    np.testing.assert_array_equal(rad.cpu().numpy(), np.maximum(r_in, 0.05))
    assert (want[:, 0] == 0).any() and (want[:, 1] == 0).any() and (want[:, 2] == H - 1).any() and (want[:, 3] == W - 1).any()
    # the re-crop: pose on the device against pose through the host (fp32 poses, as the track loop holds them)
    n = CASES[0][3]
    frames = [make_frame(sd) for sd in (CASES[0][1], CASES[1][1], CASES[0][1])]
    depth = torch.stack([torch.from_numpy(f[0].astype(np.int32)) for f in frames]).to(device)
    mask = torch.stack([torch.from_numpy(f[1]) for f in frames]).to(device)
    tr = np.stack([np.asarray(f[2], np.float64).reshape(3) for f in frames]).astype(np.float32)
    sc = np.array([CASES[0][2], CASES[1][2], 0.004], np.float32)              # (the third: fewer than 10 members -> the torch path)
    gt = {"rotation": np.stack([np.asarray(f[3]["rotation"], np.float64).reshape(3, 3) for f in frames]),
          "translation": np.stack([np.asarray(f[3]["translation"], np.float64).reshape(3) for f in frames]),
          "scale": np.array([float(np.asarray(f[3]["scale"]).reshape(-1)[0]) for f in frames])}
    np.random.seed(3)
    via_host = nocs_otf.full_data_batch_arrays(depth, mask, tr.astype(np.float64), 1.0 * sc.astype(np.float64), gt, n)
    np.random.seed(3)
    via_dev = nocs_otf.full_data_batch_arrays(depth, mask, None, None, gt, n,
                                              pose_dev=(torch.from_numpy(tr).to(device), torch.from_numpy(sc).to(device), 1.0))
    for k in ("points", "labels", "nocs"):
        assert torch.equal(via_host[k], via_dev[k])


@pytest.mark.gpu
def test_batched_recrop_rare_paths_and_golden(device):
    """A batch mixing a normal crop with one whose ball holds fewer than 10 pixels at first (radius growth: that instance
    takes the torch path) and one thinned by the permutation; each equals its own single call, and the sparse case equals
    golden G11 (the reference's own output)."""
    tag, seed, radius, n = CASES[2]
    depth, mask, center, pose = make_frame(seed)
    d, m = torch.from_numpy(depth.astype(np.int32)).to(device), torch.from_numpy(mask).to(device)
    items = [(d, m, center, radius, pose), (d, m, center, 0.004, pose), (d, m, center, 0.3, pose)]
    np.random.seed(100 + seed)
    singles = [nocs_otf.full_data_from_depth(*it, n) for it in items]
    np.random.seed(100 + seed)
    batched = nocs_otf.full_data_batch(items, n)
    for a, b in zip(singles, batched):
        _same(a, b)
    np.testing.assert_allclose(batched[0]["points"].cpu().numpy(), G[f"{tag}_points"], atol=1e-15, rtol=0)
    np.testing.assert_array_equal(batched[0]["labels"].cpu().numpy(), G[f"{tag}_labels"])


@pytest.mark.gpu
@pytest.mark.parametrize("hipgraph", [False, True])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_track_loop_otf_vs_reference_loop_golden(device, tag, hipgraph):
    """Golden G15 = the REFERENCE's own `nocs_otf=True` loop (model.py:425-452 -> full_data_from_depth_image with pre_fetched)
    under physical-regime weights at batch 1: centre / radius from the last predicted pose -> crop -> resample -> mean-subtract
    -> networks -> pose, frame after frame, free-running.  Every pose of every frame to 1e-4, every re-cropped cloud to 2e-7
    (identical pixels, identical labels), eager and with the captured step."""
    from captra_amd.configs import make_config
    from captra_amd.synthetic import OTF_LOOP_SETUPS, make_otf_trajectory, make_physical_state_dict
    from captra_amd.trainer import Trainer
    G15 = np.load(Path(__file__).resolve().parent / "golden" / "g15_otf_loop.npz")
    frames, dseed, wseed, tseed = OTF_LOOP_SETUPS[tag]
    cfg = make_config("1", experiment_dir="/tmp/captra_otf_loop_test", nocs_otf=True, hipgraph=hipgraph)
    cfg["device"] = device
    cfg["init_frame"]["gt"] = False
    trainer = Trainer(cfg)
    model = trainer.model
    model.load_state_dict(make_physical_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, wseed, 1, True, "nocs"))
    model.use_graph = hipgraph
    data = make_otf_trajectory(1, frames, seed=dseed)
    torch.manual_seed(tseed)
    np.random.seed(tseed)
    pred, _ = trainer.test(data, save=False, no_eval=True)
    assert len(pred["poses"]) == frames
    for i, pose in enumerate(pred["poses"]):
        if i > 0:
            np.testing.assert_array_equal(model.feed_dict[i]["labels"].cpu().numpy(), G15[f"{tag}_{i}_labels"], err_msg=f"labels of frame {i}")
            np.testing.assert_allclose(model.feed_dict[i]["points"].cpu().numpy(), G15[f"{tag}_{i}_points"], atol=2e-7, rtol=0, err_msg=f"cloud of frame {i}")
            np.testing.assert_allclose(model.npcs_feed_dict[i]["nocs"].cpu().numpy(), G15[f"{tag}_{i}_nocs"], atol=2e-6, rtol=0, err_msg=f"gt nocs of frame {i}")
        for key in ("rotation", "translation", "scale"):
            np.testing.assert_allclose(pose[key].cpu().numpy(), G15[f"{tag}_{i}_{key}"], atol=1e-4, rtol=0, err_msg=f"{key} of frame {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["deferred", "lanes_deferred", "every_frame_rare", "lanes_every_frame_rare"])
def test_track_loop_otf_deferred_check_replays_rare_frames(device, mode, monkeypatch):
    """The re-crop without a host round trip (captra_amd/model.py _OtfCheck: the member counts stay on the device, the verdict
    'a rare-path instance was met' is read one frame late) against the SYNCHRONOUS stage (CAPTRA_OTF_DEFER off: what golden G15
    pins) from the same start, bit for bit, in two regimes: as shipped, and with a stride bound so small that EVERY frame's lists
    outgrow it -- every frame is flagged, read a frame late and run again on the synchronous stage, the last one after the loop --
    for the single-batch loop and for the two lanes.  A replayed frame leaves nothing of its first run behind."""
    import captra_amd.model as M
    from captra_amd.configs import make_config
    from captra_amd.synthetic import OTF_LOOP_SETUPS, make_otf_trajectory, make_physical_state_dict
    from captra_amd.trainer import Trainer
    lanes = mode.startswith("lanes")
    reads = []
    orig = M._OtfCheck.read
    monkeypatch.setattr(M._OtfCheck, "read", lambda self: (reads.append(orig(self)), reads[-1])[1])
    if mode.endswith("rare"):
        monkeypatch.setattr(M, "_otf_bound", lambda longest, n: n)          # 4096: every ~15 k list is "rare"
        monkeypatch.setattr(M, "OTF_FIRST_BOUND", 1)
    tags = ["a", "b"]
    setups = [OTF_LOOP_SETUPS[t] for t in tags]
    frames, wseed = setups[0][0], setups[0][2]
    cfg = make_config("1", experiment_dir="/tmp/captra_otf_defer_test", nocs_otf=True, hipgraph=True)
    cfg["device"] = device
    cfg["init_frame"]["gt"] = True
    trainer = Trainer(cfg)
    model = trainer.model
    model.load_state_dict(make_physical_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, wseed, 1, True, "nocs"))
    reps = 16 if lanes else 1
    parts = [make_otf_trajectory(1, frames, seed=s[1]) for s in setups for _ in range(reps)] if lanes else [make_otf_trajectory(1, frames, seed=setups[0][1])]
    data = _cat_trajectories(parts) if len(parts) > 1 else parts[0]
    model.otf_lanes = lanes
    torch.manual_seed(setups[0][3]); np.random.seed(setups[0][3])
    pred, _ = trainer.test(data, save=False, no_eval=True)
    assert len(reads) >= frames - 1                        # every frame's verdict was read ...
    assert all(r for r, _ in reads) == mode.endswith("rare") and any(r for r, _ in reads) == mode.endswith("rare")   # ... and said what the regime implies
    # the same loop on the synchronous stage from the same start is the comparand (init_frame.gt: no seeded noise in the start pose)
    monkeypatch.setattr(M, "OTF_DEFER", False)
    clouds = [model.feed_dict[i]["points"].clone() for i in range(1, frames)]
    torch.manual_seed(setups[0][3]); np.random.seed(setups[0][3])
    data2 = _cat_trajectories([make_otf_trajectory(1, frames, seed=s[1]) for s in setups for _ in range(reps)]) if lanes else make_otf_trajectory(1, frames, seed=setups[0][1])
    pred2, _ = trainer.test(data2, save=False, no_eval=True)
    for i in range(1, frames):
        assert torch.equal(clouds[i - 1], model.feed_dict[i]["points"]), f"cloud of frame {i}"
        for key in ("rotation", "translation", "scale"):
            assert torch.equal(pred["poses"][i][key], pred2["poses"][i][key]), (key, i)


def _cat_trajectories(parts):
    """Concatenate single-trajectory frame lists (captra_amd.synthetic.make_otf_trajectory(1, ...)) along the batch axis."""
    out = []
    for frames in zip(*parts):
        f0 = frames[0]
        meta = {"path": sum([f["meta"]["path"] for f in frames], []), "ori_path": sum([f["meta"]["ori_path"] for f in frames], []),
                "points_mean": torch.cat([f["meta"]["points_mean"] for f in frames]), "nocs_corners": torch.cat([f["meta"]["nocs_corners"] for f in frames]),
                "pre_fetched": {k: torch.cat([f["meta"]["pre_fetched"][k] for f in frames]) for k in ("depth", "mask")},
                "nocs2camera": [{k: torch.cat([f["meta"]["nocs2camera"][p][k] for f in frames]) for k in ("rotation", "translation", "scale")}
                                for p in range(len(f0["meta"]["nocs2camera"]))]}
        out.append({"points": torch.cat([f["points"] for f in frames]), "labels": torch.cat([f["labels"] for f in frames]),
                    "nocs": torch.cat([f["nocs"] for f in frames]), "meta": meta})
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [True, False])
def test_track_loop_otf_batch32_vs_reference_loop_golden(device, lanes):
    """The 32-trajectory forms of the re-crop loop (two lanes half a frame apart, and the single batch) anchored to the
    REFERENCE: the batch is 16 copies each of golden G15's two trajectories, started from the golden's (seeded, perturbed)
    initial poses; every trajectory must reproduce the reference's batch-1 loop to 1e-4 on every frame."""
    from captra_amd.configs import make_config
    from captra_amd.synthetic import OTF_LOOP_SETUPS, make_otf_trajectory, make_physical_state_dict
    from captra_amd.trainer import Trainer
    G15 = np.load(Path(__file__).resolve().parent / "golden" / "g15_otf_loop.npz")
    T = min(OTF_LOOP_SETUPS[t][0] for t in ("a", "b"))
    assert OTF_LOOP_SETUPS["a"][2] != OTF_LOOP_SETUPS["b"][2]      # different weight seeds: one model per golden, run one after the other
    for tag in ("a", "b"):
        _, dseed, wseed, _ = OTF_LOOP_SETUPS[tag]
        cfg = make_config("1", experiment_dir="/tmp/captra_otf_b32_test", nocs_otf=True, hipgraph=True, otf_lanes=lanes)
        cfg["device"] = device
        cfg["init_frame"]["gt"] = False
        trainer = Trainer(cfg)
        model = trainer.model
        model.load_state_dict(make_physical_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, wseed, 1, True, "nocs"))
        model.use_graph = True
        model.otf_lanes = lanes
        single = make_otf_trajectory(1, T, seed=dseed)
        data = _cat_trajectories([single] * 32)
        init = {k: torch.from_numpy(np.repeat(G15[f"{tag}_0_{k}"], 32, axis=0)).to(device) for k in ("rotation", "translation", "scale")}
        model._initial_pose = lambda init=init: {k: v.clone() for k, v in init.items()}
        np.random.seed(1)
        pred, _ = trainer.test(data, save=False, no_eval=True)
        for i in range(1, T):
            for key in ("rotation", "translation", "scale"):
                got = pred["poses"][i][key].cpu().numpy()
                ref = np.repeat(G15[f"{tag}_{i}_{key}"], 32, axis=0)
                np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0, err_msg=f"{tag}: {key} of frame {i}")
