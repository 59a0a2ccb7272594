"""Instruction mix of the kernels in a hipcc -S listing: per kernel (name filter), counts by class and the most frequent opcodes.
    hipcc ... -S --cuda-device-only src.hip -o out.s ; python tools/asm_mix.py out.s [name-substring]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
    for i, (pos, name) in enumerate(starts):
        if flt not in name:
            continue
        end = starts[i + 1][0] if i + 1 < len(starts) else len(txt)
        body = txt[pos:end].split(".Lfunc_end")[0]
        c = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line[0] in ".;/" or line.endswith(":") or ":" in line.split()[0]:
                continue
            c[line.split()[0]] += 1
        grp = collections.Counter()
        for op, n in c.items():
            g = ("mfma" if op.startswith("v_mfma") else "ds" if op.startswith("ds_") else
                 "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "waitcnt" if op.startswith("s_waitcnt") else
                 "s_nop" if op.startswith("s_nop") else "salu" if op.startswith("s_") else
                 "accvgpr" if op.startswith("v_accvgpr") else "valu" if op.startswith("v_") else "other")
            grp[g] += n
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        print(short, "total", sum(c.values()), dict(grp))
        print("    ", c.most_common(16))


if __name__ == "__main__":
    main()
