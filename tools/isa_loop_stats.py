#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in a hipcc -S listing (CPU-side proxy for the issue-bound part of a
kernel): python tools/isa_loop_stats.py file.s <mangled-kernel-name-substring>"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or
                 (l.startswith("_Z") and key in l and ": ;" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            loops.append((labels[m.group(1)], i))

    def cls(op):
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "VMOV"
        if op.startswith("v_"):
            return "VALU"
        if op.startswith("s_"):
            return "SALU"
        if op.startswith("scratch_"):
            return "SCRATCH"
        if op.startswith(("global_", "buffer_", "flat_")):
            return "VMEM"
        if op.startswith("ds_"):
            return "LDS"
        return "other"

    print("kernel lines", len(body))
    for a, b in loops:
        c = Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if l and l[0] not in ";." and not l.endswith(":"):
                c[cls(l.split()[0])] += 1
        print("loop %5d..%5d  %s" % (a, b, dict(c)))


if __name__ == "__main__":
    main()
