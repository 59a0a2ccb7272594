"""Summarise rocprofv3 --pmc CSV output directories as one text table: per kernel, the mean counter
value per dispatch (and the mean dispatch duration seen in that pass).

    python tools/pmc_summary.py gpurun_out/r01_fetch gpurun_out/r01_write gpurun_out/r01_sq

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the TCC_EA request counters;
MI355X_MICROARCH.md's gfx950 note applies (FETCH_SIZE tallies 128-byte requests as 64 B: the table
prints both the raw value and the doubled, corrected one)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone.*\])?$", "", name)
    return name[:90]


def main():
    table = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, dispatches]
    argv = sys.argv[1:]
    json_out = None
    if "--json" in argv:
        i = argv.index("--json")
        json_out = argv[i + 1]
        del argv[i:i + 2]
    for d in argv:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            tag = d.rstrip("/").split("_")[-1]
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
                    per_dispatch[(row["Dispatch_Id"], f"us_in_{tag}_pass")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            for (disp, counter), v in per_dispatch.items():
                cell = table[names[disp]][counter]
                cell[0] += v
                cell[1] += 1
    counters = sorted({c for k in table.values() for c in k})
    print(f"{'kernel':92s} " + " ".join(f"{c:>26s}" for c in counters))
    for kern in sorted(table, key=lambda k: -max((v[0] for v in table[k].values()), default=0)):
        cells = []
        for c in counters:
            s, n = table[kern].get(c, [0.0, 0])
            cells.append(f"{(s / n if n else float('nan')):26.1f}")
        print(f"{kern:92s} " + " ".join(cells))
    if json_out:
        with open(json_out, "w") as fh:
            import os
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import csrc_fingerprint      # identity of the kernel sources these counters were collected on
            json.dump({"csrc_sha16": csrc_fingerprint(), "unit_note": "mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: double FETCH_SIZE)",
                       "kernels": {k: {c: {"mean": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items() if v[1]} for k, cs in table.items()}},
                      fh, indent=1)
    print("\n# values are the MEAN PER DISPATCH (summed over XCDs/SEs).  FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them;")
    print("# per MI355X_MICROARCH.md, on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x: corrected bytes = 2 * FETCH_SIZE * 1024.")


if __name__ == "__main__":
    main()
