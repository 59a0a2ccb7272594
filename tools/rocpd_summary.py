"""Summarise a rocprofv3 rocpd database (kernel trace) as text.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--step-trace N]

Prints the per-kernel statistics table (calls, total, average, min, max, share — what
`rocprofv3 --stats` reports) and, with --step-trace, the dispatch sequence of the N-th
`canonicalize` -> next-`part_fit_st` window (one tracking step) with launch gaps.
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)   # drop the parameter list
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--step-trace", type=int, default=None)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
    stats = {}
    for name, s, e, *_ in rows:
        d = (e - s) / 1e3
        st = stats.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0.0
    print(f"# kernel dispatches: {len(rows)}   sum of kernel time: {total / 1e3:.3f} ms   first-to-last span: {span / 1e3:.3f} ms")
    print(f"{'kernel':112s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[: args.top]:
        print(f"{name:112s} {v[0]:6d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {v[2]:10.2f} {v[3]:10.2f} {100 * v[1] / total:6.2f}")
    if args.step_trace is not None:
        starts = [i for i, r in enumerate(rows) if "canonicalize_kernel" in r[0]]
        # a step has two canonicalise launches (CoordNet, RotationNet): windows start at every 2nd
        i0 = starts[2 * args.step_trace]
        i1 = starts[2 * args.step_trace + 2] if 2 * args.step_trace + 2 < len(starts) else len(rows)
        print(f"\n# dispatch sequence of step {args.step_trace}: {i1 - i0} dispatches, "
              f"{(rows[i1 - 1][2] - rows[i0][1]) / 1e6:.3f} ms wall, {sum((r[2] - r[1]) for r in rows[i0:i1]) / 1e6:.3f} ms in kernels")
        prev_end = rows[i0][1]
        print(f"{'gap_us':>8s} {'dur_us':>9s} {'grid(blocks)':>16s} {'wg':>5s} {'vgpr':>5s} {'lds':>7s}  kernel")
        for name, s, e, gx, gy, gz, wx, vg, lds in rows[i0:i1]:
            blocks = f"{gx // max(wx, 1)}x{gy}x{gz}"
            print(f"{(s - prev_end) / 1e3:8.1f} {(e - s) / 1e3:9.1f} {blocks:>16s} {wx:5d} {vg:5d} {lds:7d}  {short(name)[:90]}")
            prev_end = e


if __name__ == "__main__":
    main()
