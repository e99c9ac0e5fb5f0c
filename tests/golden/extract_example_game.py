#!/usr/bin/env python3
"""Extract the seeded example game of the reference's log viewer (log-viewer/index.example.html:10-264): the only
whole-game vector in the reference with its seed, every draw, hora deltas and ura markers.  Per-decision `meta` objects
(q-values of the engine that played it) are dropped; only the mjai events are kept.
Output: tests/golden/example_game.jsonl (one compact JSON object per line, keys in the reference's serialisation order).
"""
import json
import os
import re

SRC = "/root/reference/log-viewer/index.example.html"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_game.jsonl")

html = open(SRC).read()
block = re.search(r"allActions = `\n(.*?)\n\s*`\.trim\(\)", html, re.S).group(1)
lines = []
for l in block.split("\n"):
    ev = json.loads(l)
    ev.pop("meta", None)
    lines.append(json.dumps(ev, separators=(",", ":")))
with open(OUT, "w") as f:
    f.write("\n".join(lines) + "\n")
print(len(lines), "events ->", OUT)
