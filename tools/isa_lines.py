#!/usr/bin/env python3
"""Static per-source-line instruction histogram of one kernel from `hipcc -S -gline-tables-only` output.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -gline-tables-only -S -o k.s mortal_amd/csrc/mj_capi.hip
  python tools/isa_lines.py k.s _Z7mj_k_sp8SpParams [pattern]

Prints, per (file, line), the number of ISA instructions attributed to it (and how many match `pattern`, e.g. scratch_).
Static counts, not execution counts: use them to compare two builds of the same loop body.
"""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    files, cur, on = {}, None, False
    hist, hits = collections.Counter(), collections.Counter()
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if s.startswith(sym + ":"):
            on = True
            continue
        if on and s.startswith(".Lfunc_end"):
            break
        if not on:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        hist[cur] += 1
        if pat and pat.search(s):
            hits[cur] += 1
    tot = sum(hist.values())
    print("total", tot)
    for (k, v) in sorted(hist.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
        print(f"{k[0]}:{k[1]}\t{v}\t{hits.get(k, 0) if pat else ''}")


if __name__ == "__main__":
    main()
