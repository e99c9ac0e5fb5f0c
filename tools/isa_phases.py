#!/usr/bin/env python3
"""Static instruction count of one function split by source-line ranges of ONE file (inlined callees are charged to the
call site's range): python tools/isa_phases.py k.s <symbol> <file> name:first_line ...   (k.s built with -gline-tables-only)"""
import collections
import re
import sys

path, sym, fname = sys.argv[1:4]
marks = sorted((int(x.split(":")[1]), x.split(":")[0]) for x in sys.argv[4:])
files, on, cur = {}, False, None
cnt = collections.Counter()
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
        continue
    if s.startswith(sym) and ":" in s[:len(sym) + 80] and not on:
        on = True
        continue
    if on and s.startswith(".Lfunc_end"):
        break
    if not on:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        if files.get(int(m.group(1))) == fname and int(m.group(2)) > 0:
            line = int(m.group(2))
            cur = None
            for first, name in marks:
                if line >= first:
                    cur = name
        continue
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    cnt[cur] += 1
print("total", sum(cnt.values()))
for k, v in cnt.items():
    print(f"{k}\t{v}")
