"""Multi-process cover of the N>1 path on CPU: gloo backend, world_size 2 (runs without a GPU)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from captra_amd.parallel import POSE_RECORD, PoseExchange, pack_pose, shard_range, unpack_pose


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(total, world, r)]
            assert got == list(range(total))
            sizes = [len(shard_range(total, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _fake_pose(b, p, seed):
    g = torch.Generator().manual_seed(seed)
    return {"rotation": torch.randn(b, p, 3, 3, generator=g), "translation": torch.randn(b, p, 3, 1, generator=g),
            "scale": torch.rand(b, p, generator=g) + 0.5}


def test_pack_unpack_roundtrip():
    pose = _fake_pose(5, 4, 0)
    valid = torch.tensor([[True, False, True, True]] * 5)
    rec = pack_pose(pose, valid)
    assert rec.shape == (5, 4, POSE_RECORD)
    back, v = unpack_pose(rec)
    for k in pose:
        assert torch.equal(back[k], pose[k])
    assert torch.equal(v, valid)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, P = 3, 2
        ex = PoseExchange(B, P, "cpu", world, rank)
        for frame in range(3):                       # a few frames, sync and async flavours
            pose = _fake_pose(B, P, 100 * frame + rank)
            out = ex.all_gather(pose, async_op=(frame % 2 == 1))
            out = ex.wait()
            for r in range(world):
                exp = pack_pose(_fake_pose(B, P, 100 * frame + r))
                assert torch.equal(out[r * B:(r + 1) * B], exp), (frame, r)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_pose_all_gather_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


def test_trajectory_npz_roundtrip(tmp_path):
    """captra_amd/trajectory_io.py: frame dicts -> one .npz per trajectory -> batched frame dicts, unchanged."""
    import numpy as np
    import torch
    from captra_amd.trajectory_io import load_trajectory_npz, save_trajectory_npz, stack_trajectories
    from tests import clouds
    frames = clouds.make_trajectory("arti", 3, 3, seed=1)
    for b in range(3):
        save_trajectory_npz(str(tmp_path / f"t{b}.npz"), frames, b)
    back = stack_trajectories([load_trajectory_npz(str(tmp_path / f"t{b}.npz")) for b in range(3)])
    assert len(back) == len(frames)
    for f0, f1 in zip(frames, back):
        for k in ("points", "labels", "nocs"):
            assert torch.equal(f0[k], f1[k]) and f0[k].dtype == f1[k].dtype
        assert f0["meta"]["path"] == f1["meta"]["path"]
        assert torch.equal(f0["meta"]["points_mean"], f1["meta"]["points_mean"])
        assert torch.equal(f0["meta"]["nocs_corners"], f1["meta"]["nocs_corners"])
        for p0, p1 in zip(f0["meta"]["nocs2camera"], f1["meta"]["nocs2camera"]):
            for k in ("rotation", "translation", "scale"):
                np.testing.assert_array_equal(p0[k].numpy(), p1[k].numpy())


def _grad_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from captra_amd.parallel import allreduce_gradients
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = [(7, 3), (5,), (2, 2, 2), (1,), (300,)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes] + [torch.nn.Parameter(torch.zeros(3))]   # last: no grad
        for i, (p, s) in enumerate(zip(params, shapes)):
            p.grad = torch.full(s, float(rank + 1)) * (i + 1) + torch.arange(p.numel(), dtype=torch.float32).reshape(s)
        allreduce_gradients(params, world, bucket_bytes=64)          # tiny buckets: several collectives, ragged packing
        for i, (p, s) in enumerate(zip(params, shapes)):
            exp = torch.full(s, sum(r + 1 for r in range(world)) / world) * (i + 1) + torch.arange(p.numel(), dtype=torch.float32).reshape(s)
            assert torch.allclose(p.grad, exp), (i, p.grad, exp)
        assert params[-1].grad is None
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """The data-parallel training step's gradient exchange (bucketed flat all-reduce + average) on 2 CPU ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs) and dict(ret) == {0: True, 1: True}


def _run_track_world(tmp_path, world, tag, env_extra=None, extra_args=()):
    """Launch tests/track_dist_worker.py as `world` ranks (torch.distributed.run on 127.0.0.1) or as a plain process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    worker = os.path.join(root, "tests", "track_dist_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, str(tmp_path), tag, *extra_args]
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), worker, str(tmp_path), tag, *extra_args]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout[-3000:] + "\n" + res.stderr[-3000:]
    return res.stdout


def _compare_track_worlds(tmp_path, tags, n_traj=5):
    import json
    import pickle
    import numpy as np

    def same(a, b, where):
        assert type(a) is type(b), (where, type(a), type(b))
        if isinstance(a, dict):
            assert sorted(a) == sorted(b), where
            for k in a:
                same(a[k], b[k], f"{where}/{k}")
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b), where
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{where}[{i}]")
        elif isinstance(a, torch.Tensor):
            assert a.dtype == b.dtype and torch.equal(a, b), where
        elif isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and np.array_equal(a, b), where
        else:
            assert a == b, where

    base = sorted((tmp_path / tags[0] / "results" / "data").glob("*.pkl"))
    assert len(base) == n_traj                          # one pickle per trajectory
    res0 = json.load(open(tmp_path / f"{tags[0]}_result.json"))
    for tag in tags[1:]:
        other = sorted((tmp_path / tag / "results" / "data").glob("*.pkl"))
        assert [p.name for p in other] == [p.name for p in base]
        for p0, p1 in zip(base, other):
            with open(p0, "rb") as f0, open(p1, "rb") as f1:
                r0, r1 = pickle.load(f0), pickle.load(f1)
            # 'frame_nums' holds, per frame, the frame numbers of the WHOLE batch the trajectory was tracked in (the
            # reference's get_ith_from_batch passes lists of strings through, utils.py:155-172): it follows the batching,
            # which differs between worlds; every other entry is the trajectory's own and must not change by a bit
            n0, n1 = r0.pop("frame_nums"), r1.pop("frame_nums")
            assert len(n0) == len(n1) and all(set(b) <= set(a) or set(a) <= set(b) for a, b in zip(n0, n1))
            same(r0, r1, p0.name)
        res = json.load(open(tmp_path / f"{tag}_result.json"))
        assert res["frames"] == res0["frames"] == 4 * n_traj
        assert sorted(res["loss"]) == sorted(res0["loss"]) and res["loss"]
        for k in res0["loss"]:
            assert abs(res["loss"][k] - res0["loss"][k]) <= 1e-6 * max(1.0, abs(res0["loss"][k])), k
    return res0


def test_track_harness_shards_trajectories_over_ranks_gloo(tmp_path):
    """captra_amd.track under torch.distributed.run with 2 and 3 ranks (gloo, CPU): 5 synthetic trajectories shard 3+2 /
    2+2+1 (a rank plays an empty round), every frame's pose records are all-gathered, rank 0 writes all five result
    pickles -- bit for bit the world-1 run's, poses taken from the gathered records.  The networks' GPU step is replaced
    by host arithmetic (tests/track_dist_worker.py); the loop, the hooks and the harness are the product's."""
    env = {"CAPTRA_TEST_HOST_STEP": "1", "CAPTRA_DIST_BACKEND": "gloo", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""}
    _run_track_world(tmp_path, 1, "w1", env)
    out2 = _run_track_world(tmp_path, 2, "w2", env)
    _run_track_world(tmp_path, 3, "w3", env)
    assert "rank 0 of 2" in out2
    res = _compare_track_worlds(tmp_path, ["w1", "w2", "w3"])
    assert any(k.startswith("avg_pred/") for k in res["loss"])


@pytest.mark.parametrize("n_traj", [5, 19])
def test_track_harness_world8_gloo(tmp_path, n_traj):
    """The 8-rank shape of one node, blind: captra_amd.track under torch.distributed.run with EIGHT ranks (gloo,
    CPU).  5 trajectories: five ranks hold one each and three are EMPTY (they still take part in every frame's all-gather and in the
    result gathering); 19 trajectories: uneven shards 3,3,3,2,2,2,2,2 with batch_size 2 (a rank's last round is short).  Rank 0's
    result pickles are the world-1 run's bit for bit."""
    from captra_amd.parallel import shard_range
    sizes = [len(shard_range(n_traj, 8, r)) for r in range(8)]
    assert sum(sizes) == n_traj and (0 in sizes) == (n_traj == 5) and max(sizes) - min(sizes) == 1
    env = {"CAPTRA_TEST_HOST_STEP": "1", "CAPTRA_DIST_BACKEND": "gloo", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": "",
           "CAPTRA_TEST_NUM_TRAJ": str(n_traj), "OMP_NUM_THREADS": "1"}
    _run_track_world(tmp_path, 1, "w1", env)
    out8 = _run_track_world(tmp_path, 8, "w8", env)
    assert "rank 0 of 8" in out8
    _compare_track_worlds(tmp_path, ["w1", "w8"], n_traj)


def test_bench_dry_run_prints_the_launch_plan_without_a_gpu():
    """`python bench.py --gpus 8 --dry-run`: the exact launcher command of the self-spawn, the rank -> device map and the
    collective, as JSON, with no GPU touched (this container has none: it would refuse the real launch and says so)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "7"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    plan = json.loads(res.stdout.strip().splitlines()[-1])
    assert plan["gpus_requested"] == 8 and len(plan["ranks"]) == 8
    assert [r["device"] for r in plan["ranks"]] == [f"cuda:{i}" for i in range(8)]
    assert "torch.distributed.run" in plan["command"] and "--nproc-per-node=8" in plan["command"] and "--master-addr 127.0.0.1" in plan["command"]
    assert "--steps 7" in plan["command"] and "--dry-run" not in plan["command"]
    assert "nccl" in plan["collective"] and plan["scaling"] == "weak"
    import torch
    assert plan["would_refuse"] == (torch.cuda.device_count() < 8 if torch.cuda.is_available() else True)


def test_rank_cpu_binding_helpers():
    """captra_amd.parallel: the sysfs cpulist parser and the split of a GPU's NUMA-local cores among the ranks that share them."""
    from captra_amd.parallel import cpu_slice, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    cpus = list(range(96))
    parts = [cpu_slice(cpus, r, 8) for r in range(8)]
    assert all(len(p) == 12 for p in parts) and sorted(sum(parts, [])) == cpus            # even, disjoint, complete
    # four GPUs per socket: ranks 4..7 share the second socket's cores
    second = list(range(96, 192))
    parts = [cpu_slice(second, r, 8, sharers=[4, 5, 6, 7]) for r in (4, 5, 6, 7)]
    assert all(len(p) == 24 for p in parts) and sorted(sum(parts, [])) == second
    assert cpu_slice(second, 0, 8, sharers=[4, 5, 6, 7]) == second                         # not a sharer: left alone
    assert cpu_slice([5], 3, 8) == [5] and cpu_slice([], 0, 2) == []                      # fewer cores than ranks: never empty


def test_scale_tool_dry_run_lists_the_north_star_table():
    """tools/scale.py --dry-run: configs[1] at 1 / 2 / 4 / 8 ranks, configs[2] (mix6, bf16) and configs[4] at 8, each with the
    exact command and (for bench.py runs) the launch plan -- no GPU touched."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "scale.py"), "--dry-run"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    plan = json.loads(res.stdout.strip().splitlines()[-1])
    names = [r["name"] for r in plan["runs"]]
    assert names[:4] == [f"configs[1] bottle fp32 x{n}" for n in (1, 2, 4, 8)]
    assert any("mix6 bf16 x8" in n for n in names) and any("backbone16k x8" in n for n in names)
    by = {r["name"]: r for r in plan["runs"]}
    p8 = by["configs[1] bottle fp32 x8"]["launch_plan"]
    assert p8["gpus_requested"] == 8 and "nccl" in p8["collective"] and "cpu_binding" in p8
    assert "--category mix6 --mlp-dtype bf16" in by["configs[2] mix6 bf16 x8"]["command"]
    # the mix6 plan serves all six rigid categories: rank r tracks category 1 + r mod 6
    mix = by["configs[2] mix6 bf16 x8"]["launch_plan"]
    cats = [r["category"] for r in mix["ranks"]]
    assert cats == [str(1 + r % 6) for r in range(8)] and set(cats) == {"1", "2", "3", "4", "5", "6"}
    assert "--nproc-per-node=8" in by["configs[4] backbone16k x8"]["command"]
