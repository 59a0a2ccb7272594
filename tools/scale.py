#!/usr/bin/env python
"""north_star's table in one command, for a node with 8 MI355X (none has been available to this repository: the driver runs the
scaling bench itself; this is the same thing for a maintainer).

    python tools/scale.py            # runs everything below, prints one JSON line per run and a table at the end
    python tools/scale.py --dry-run  # the plan (commands, ranks, collective) as JSON, no GPU touched

Runs, each through `bench.py`'s own self-spawn (one rank per GPU, torch.distributed.run on 127.0.0.1, RCCL all-gather of the
per-frame pose records, per-rank CPU binding):
  configs[1]  bottle, fp32, 32 trajectories per GPU at 1 / 2 / 4 / 8 GPUs            (the metric's configuration; weak scaling)
  configs[2]  six rigid categories (rank r serves category 1 + r mod 6), bf16 MFMA operands, 8 GPUs x 32 = 256 frames per step
  configs[4]  16384-point clouds, 3-level set abstraction, 8 clouds per GPU x 8 = 64 (replicas: nothing to exchange)
Every line carries `rccl_world_size` as an actual all-gather saw it; scaling efficiency is value(N) / (N value(1)).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def plan(max_gpus: int = 8, steps: int = 20, warmup: int = 20) -> list[dict]:
    legs = ["--no-otf", "--no-b1", "--no-legs", "--no-cpu-baseline"]
    runs = []
    n = 1
    while n <= max_gpus:
        runs.append({"name": f"configs[1] bottle fp32 x{n}", "gpus": n,
                     "cmd": [sys.executable, BENCH, "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + (legs if n > 1 else ["--no-legs"])})
        n *= 2
    runs.append({"name": f"configs[2] mix6 bf16 x{max_gpus}", "gpus": max_gpus,
                 "cmd": [sys.executable, BENCH, "--gpus", str(max_gpus), "--steps", str(steps), "--warmup", str(warmup), "--category", "mix6",
                         "--mlp-dtype", "bf16"] + legs})
    runs.append({"name": f"configs[4] backbone16k x{max_gpus}", "gpus": max_gpus,
                 "cmd": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={max_gpus}", "--master-addr", "127.0.0.1",
                         "--master-port", "29517", os.path.join(ROOT, "tools", "bench_backbone.py"), "--gpus", str(max_gpus), "--npoint", "2048", "512",
                         "--steps", str(steps), "--warmup", "10"] if max_gpus > 1 else
                        [sys.executable, os.path.join(ROOT, "tools", "bench_backbone.py"), "--npoint", "2048", "512", "--steps", str(steps), "--warmup", "10"]})
    return runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args()
    runs = plan(a.max_gpus, a.steps, a.warmup)
    if a.dry_run:
        out = []
        for r in runs:
            entry = {"name": r["name"], "gpus": r["gpus"], "command": " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in r["cmd"][1:])}
            if r["cmd"][1] == BENCH:
                res = subprocess.run(r["cmd"] + ["--dry-run"], capture_output=True, text=True)
                entry["launch_plan"] = json.loads(res.stdout.strip().splitlines()[-1]) if res.returncode == 0 else {"error": res.stderr[-300:]}
            out.append(entry)
        print(json.dumps({"runs": out, "efficiency": "value(N) / (N * value(1)) over the configs[1] runs"}))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rows = []
    for r in runs:
        res = subprocess.run(r["cmd"], capture_output=True, text=True, env=env, cwd=ROOT)
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not lines:
            rows.append((r["name"], r["gpus"], None, None, (res.stderr or res.stdout)[-200:].replace("\n", " ")))
            continue
        d = json.loads(lines[-1])
        print(lines[-1], flush=True)
        rows.append((r["name"], r["gpus"], d["value"], d.get("unit"), f"rccl_world_size {d.get('rccl_world_size')}, {d.get('ms_per_step')} ms/step"))
    base = next((v for n, g, v, *_ in rows if g == 1 and v), None)
    print(f"{'run':34s} {'GPUs':>4s} {'value':>12s} {'unit':>10s} {'efficiency':>10s}  notes")
    for name, g, v, unit, note in rows:
        eff = f"{v / (g * base):.3f}" if (v and base and name.startswith("configs[1]")) else ""
        print(f"{name:34s} {g:4d} {v if v is not None else 'FAILED':>12} {unit or '':>10s} {eff:>10s}  {note}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
