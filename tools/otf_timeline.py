"""Timeline check of the nocs_otf lanes: run the 32-trajectory loop under `rocprofv3 --kernel-trace --output-format csv` and
report how much of the re-crop sampler's time (fps kernels) overlaps the other lane's MFMA kernels.
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/otf_timeline.py run
    python tools/otf_timeline.py report OUT"""
import csv
import glob
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))


def run():
    import bench_otf
    bench_otf.track_loop(32, frames=8, configs=((True, True),))


def report(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    t0 = rows[0][0]
    fps = [(s, e) for s, e, n, *_ in rows if "fps" in n and e - s > 500_000]
    mf = [(s, e) for s, e, n, *_ in rows if ("sa_wave" in n or "pw_direct" in n) and e - s > 100_000]
    print(f"{len(rows)} dispatches, {len(fps)} long sampler launches, {len(mf)} long MFMA launches")
    tot = ov = 0
    for s, e in fps[-12:]:
        o = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in mf)
        tot += e - s
        ov += o
        print(f"  sampler {1e-6 * (s - t0):9.3f} .. {1e-6 * (e - t0):9.3f} ms  ({1e-6 * (e - s):.3f} ms), MFMA kernels running during it: {1e-6 * o:.3f} ms")
    print(f"overlap {ov / max(tot, 1):.2f} of the sampler time")
    qs = {}
    for s, e, n, q, st in rows[-400:]:
        qs.setdefault((q, st), []).append(n[:30])
    for k, v in qs.items():
        print("queue/stream", k, len(v), sorted(set(v))[:6])


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
