#!/usr/bin/env python3
"""Extract the event stream of the reference's "encode obs" criterion benchmark (libriichi/benches/bench.rs:136-241): a
full-information kyoku of ~100 events that PlayerState(3) replays before `encode_obs(4, false)` is timed.  The reference
holds no expected output for it; it serves as one more reference-authored input for the oracle-vs-device obs comparison.
Only the test VECTOR (the JSON lines) is extracted.  Output: tests/golden/bench_kyoku.jsonl
"""
import json
import os
import re

SRC = "/root/reference/libriichi/benches/bench.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_kyoku.jsonl")

src = open(SRC).read()
m = re.search(r'fn encode_obs\(c: &mut Criterion\) \{\s*let log = r#"(.*?)"#;', src, re.S)
lines = [l.strip() for l in m.group(1).strip().split("\n") if l.strip()]
for l in lines:
    json.loads(l)
with open(OUT, "w") as f:
    f.write("\n".join(lines) + "\n")
print(len(lines), "events ->", OUT)
