#!/usr/bin/env python3
"""Extract the mjai event lines of the reference's serde round-trip test (mjai/event.rs:260-294, `json_consistency`)
into tests/golden/event_lines.jsonl (run in the build container only).  Only the test vectors are taken."""
import json
import os
import re

SRC = "/root/reference/libriichi/src/mjai/event.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "event_lines.jsonl")

text = open(SRC).read()
body = text[text.index("fn json_consistency"):]
block = re.search(r'r#"(.*?)"#', body, re.S).group(1)
lines = [l.strip() for l in block.strip().splitlines() if l.strip()]
for l in lines:
    json.loads(l)
open(OUT, "w").write("\n".join(lines) + "\n")
print(len(lines), "lines ->", OUT)
