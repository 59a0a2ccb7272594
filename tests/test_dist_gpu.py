"""The N > 1 path on real kernels.  A gpurun box has ONE GPU, so these tests start two ranks that share it over gloo
(CAPTRA_DIST_BACKEND=gloo / CAPTRA_BENCH_SHARE_GPU=1 -- functional modes, never a measurement): everything but the
collective library is the product path an 8-GPU node runs (sharding, per-frame pose all-gather, rank-0 result files,
bench.py's self-spawn, per-rank timing, world-size report)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.test_parallel_cpu import _compare_track_worlds, _run_track_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_track_harness_two_ranks_equal_single_process_bit_for_bit(device, tmp_path):
    """`python -m captra_amd.track` as 2 ranks (3 + 2 trajectories, the second round is short on rank 0 and EMPTY on rank 1)
    against the single-process run: the five result pickles are identical bit for bit (the poses in the 2-rank files are
    the all-gathered per-frame records), frames and trajectory-weighted errors agree."""
    env = {"CAPTRA_DIST_BACKEND": "gloo"}
    _run_track_world(tmp_path, 1, "g1", env)
    out = _run_track_world(tmp_path, 2, "g2", env)
    assert "rank 0 of 2" in out
    _compare_track_worlds(tmp_path, ["g1", "g2"])
    _run_track_world(tmp_path, 2, "g2graph", env, extra_args=("--hipgraph",))
    _compare_track_worlds(tmp_path, ["g1", "g2graph"])


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_n_without_a_launcher_spawns_n_ranks(device):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself (torch.distributed.run on 127.0.0.1), never
    a silent single-GPU run: the line reports n_gpus 2, the world size seen by an actual all-gather, and one row of block
    times per rank.  (Two ranks share this box's GPU over gloo here: functional test mode, flagged in the line.)"""
    res = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--min-warmup", "1", "--repeats", "2", "--min-timed-s", "0", "--batch", "4",
                  "--no-cpu-baseline", "--no-kernel-timing"], {"CAPTRA_BENCH_SHARE_GPU": "1"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, res.stdout          # rank 0 prints ONE JSON line
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["timed_blocks"]["n"] == 2 and out["config"]["trajectories_per_gpu"] == 4
    assert "gloo" in out["collective_backend"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 4 * 2 / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 1e-2
    assert out["pose_match"]["within_1e-4"] and out["pose_match"]["agree_5deg5cm"] == 1.0


def test_bench_refuses_more_ranks_than_gpus(device):
    """More ranks than GPUs is refused loudly (exit code != 0, message), both for the self-spawn and under a launcher."""
    n = torch.cuda.device_count() + 1
    res = _bench(["--gpus", str(n), "--steps", "1"])
    assert res.returncode != 0 and "GPU(s)" in (res.stdout + res.stderr)
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(os.environ, **env),
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stdout + res.stderr)


def test_bench_line_contract_single_gpu(device):
    """The default command's JSON line (shortened: 2 steps, 2 blocks, small CPU budget) carries what the driver and the judge
    read: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
    data / config.workload, `roofline` {bound, achieved, peak, unit, frac, traffic} on the MFMA family, `cpu_baseline`
    {value, unit, cores, kind, sample}, the accuracy of the timed trajectories (`pose_match`) and the `otf` leg."""
    res = _bench(["--steps", "2", "--warmup", "1", "--min-warmup", "2", "--repeats", "2", "--min-timed-s", "0", "--cpu-budget", "3"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, res.stdout
    out = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "pose_match", "otf", "hbm_ops"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - out["config"]["trajectories_per_gpu"] / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0)
    # measured in the run itself where rocprofv3 is there (two counter passes over a child), else the committed profile's figure
    assert r["traffic_source"].startswith("measured in this run") or "traffic_live" in r
    if "traffic_committed_profile" in r and r["traffic_committed_profile"]:
        assert 0.7 < r["traffic"] / r["traffic_committed_profile"] < 1.4
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == out["unit"] and c["sample"]
    assert out["pose_match"]["within_1e-4"] and out["pose_match"]["agree_5deg5cm"] == 1.0
    assert out["otf"]["single_batch"]["value"] > 0 and out["otf"]["two_lanes"]["value"] > 0
    b1 = out["b1"]                     # single-trajectory latency (the reference's own measuring convention, README.md:267)
    assert 0.3 < b1["pre_cropped"]["ms_per_frame"] < 5.0 and b1["pre_cropped"]["ms_per_frame"] < b1["nocs_otf"]["ms_per_frame"] < 20.0
    h = out["hbm_ops"]
    assert h["per_level"]["frac"] > h["frac"] and h["fill_probe_GB/s"] > 1000 and "equivalent_frac" not in h
    # the reference's QueryAndGroup module, one launch per call, and the summary where the driver's parser keeps it
    q = h["query_and_group"]
    assert q["launches"] == 10 and q["frac"] > 0.3 and abs(q["frac"] - r["hbm_ops"]["query_and_group_frac"]) < 1e-9
    assert r["hbm_ops"]["frac"] == h["frac"] and r["hbm_ops"]["per_level_frac"] == h["per_level"]["frac"]
    assert len(line[0]) < 8000, f"the bench line must fit the driver's 8 KB tail: {len(line[0])} bytes"


def test_build_then_smoke_in_one_process(device):
    """`__graft_entry__.build()` followed by `smoke()` in the SAME process (a driver may do that on the GPU box): build() must not load
    libcaptra_hip.so before torch -- that leaves the process with two HIP runtimes and every launch fails with hipErrorNoDevice."""
    res = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "smoke ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


def test_rccl_communicator_world1_graph_lanes_and_exchange(device):
    """What a 1-GPU box can exercise of the RCCL path, in a process of its own (tools/check_dist_graph.py):
    init_process_group("nccl", world_size=1, device_id=...), the step captured AFTER the communicator is up, five replays and
    twelve free-running lane steps each followed by bench.py's asynchronous all_gather_into_tensor (waited for one step
    later), the product harness's FramePoseGather incl. its short-batch padding, barrier, clean destroy_process_group."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dist_graph.py")], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    for line in ("ok: captured with overlap_nets", "ok: free-running lanes + all-gather", "ok: track harness frame exchange",
                 "ok: process group destroyed"):
        assert line in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_bench_under_a_launcher_runs_its_exchange_through_rccl_at_world1(device):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with CAPTRA_BENCH_FORCE_DIST=1: the very code an 8-GPU
    launch runs per rank -- init_process_group("nccl", device_id), async per-step all-gather, barrier inside the timed
    blocks' sync, the world-size all-gather, destroy -- with the one rank this box has."""
    env = dict(os.environ, CAPTRA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--min-warmup", "1", "--repeats", "2", "--min-timed-s", "0", "--batch", "32",
           "--no-cpu-baseline", "--no-kernel-timing", "--no-otf"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["collective_backend"] == "nccl (RCCL)" and out["rccl_world_size"] == 1 and out["n_gpus"] == 1
    assert "2 free-running lanes" in out["config"]["launch"] and out["pose_match"]["within_1e-4"]


def test_pose_exchange_packs_on_the_device_like_pack_pose(device):
    """PoseExchange.all_gather on CUDA tensors packs the records with one kernel (captra_pack_pose) — the same (B,P,14)
    records as parallel.pack_pose, with and without a validity mask, also for non-contiguous inputs (torch path)."""
    from captra_amd.parallel import PoseExchange, pack_pose, unpack_pose
    g = torch.Generator().manual_seed(3)
    B, P = 5, 3
    pose = {"rotation": torch.randn(B, P, 3, 3, generator=g).to(device), "translation": torch.randn(B, P, 3, 1, generator=g).to(device),
            "scale": torch.rand(B, P, generator=g).to(device)}
    valid = (torch.rand(B, P, generator=g) > 0.4).float().to(device)
    ex = PoseExchange(B, P, device)
    assert torch.equal(ex.all_gather(pose).clone(), pack_pose(pose))
    assert torch.equal(ex.local, pack_pose(pose))
    assert torch.equal(ex.all_gather(pose, valid).clone(), pack_pose(pose, valid))
    back, ok = unpack_pose(ex.gathered)
    assert torch.equal(back["rotation"], pose["rotation"]) and torch.equal(ok, valid > 0.5)
    strided = dict(pose, rotation=pose["rotation"].transpose(2, 3))
    assert torch.equal(ex.all_gather(strided).clone(), pack_pose(strided))


def test_headline_is_stable_across_fresh_processes(device):
    """which hardware queues the lanes' streams get is decided when a process creates them, and the
    pre-cropped lanes keep the one-graph form (captra_amd/graph.py) -- five fresh processes of the headline command (short timed
    region) must agree within 5 % and none may report a bimodal block distribution."""
    vals, bimodal = [], []
    for _ in range(5):
        res = _bench(["--leg", "--steps", "20", "--repeats", "5", "--min-timed-s", "1.5", "--no-kernel-timing"])
        assert res.returncode == 0, res.stderr[-2000:]
        line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        vals.append(line["ms_per_step"])
        bimodal.append(line["timed_blocks"]["bimodal"])
    assert max(vals) <= 1.05 * min(vals), vals
    assert not any(bimodal), (vals, bimodal)
