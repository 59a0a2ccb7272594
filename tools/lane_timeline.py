"""Timeline of the default bench command (two free-running lanes, two networks side by side): which kernels run when, per queue.
    cd /tmp; rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py --steps 5 --warmup 2 --repeats 1 --min-timed-s 0 --min-warmup 2 --no-pose-match --no-cpu-baseline --no-kernel-timing --no-otf --no-b1 [--mlp-dtype bf16]
    python tools/lane_timeline.py OUT [window_ms]
Prints the last `window_ms` of dispatches (start, duration, queue, kernel) and the busy fraction per kernel family."""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:58]


def main():
    d = sys.argv[1]
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    fits = [r for r in rows if "part_fit" in r[2]]
    t1 = fits[-3][1] if len(fits) >= 3 else rows[-1][1]          # a window inside the timed steps
    sel = [r for r in rows if t1 - win * 1e6 <= r[0] <= t1]
    t0 = sel[0][0]
    qids = sorted({r[3] for r in sel})
    print("queues:", qids)
    for s, e, n, q, g, w in sel:
        print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f}  q{qids.index(q)}  {g // max(w, 1):6d} wg  {n}")
    # how many kernels run concurrently, time-weighted
    ev = sorted([(s, 1) for s, *_ in sel] + [(e, -1) for _, e, *_ in sel])
    lvl, last, hist = 0, ev[0][0], {}
    for t, dlt in ev:
        hist[lvl] = hist.get(lvl, 0) + (t - last)
        lvl += dlt
        last = t
    tot = sum(hist.values())
    print("concurrency histogram (kernels in flight: share of the window):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main()
