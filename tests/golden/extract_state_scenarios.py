#!/usr/bin/env python3
"""Extract the reference's PlayerState scenario tests (libriichi/src/state/test.rs:71-1418) into a JSON fixture.

Runs in the build container only.  Each `#[test] fn` becomes an ordered list of steps:
  {"new": player_id}                      PlayerState::new / from_log
  {"ev": {...mjai event...}, "bind": "cans"|null}
  {"set_tehai": "hand string", "len_div3": n}   direct field pokes used by the first tests
  {"call": "update_waits_and_furiten" | "set_can_chi_from_tile", "arg": tile}
  {"clone": "name"} / {"use": "name"}     ps.clone() handling of double_chankan_ron
  {"assert": "<the Rust assertion text>"} evaluated by tests/test_oracle_state.py (pattern table there)
Only test VECTORS (events, hands, expected values) are extracted; no reference code is copied.
Output: tests/golden/state_scenarios.json
"""
import json
import os
import re

SRC = "/root/reference/libriichi/src/state/test.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_scenarios.json")

TILE_RE = r"t!\(([^)]+)\)"


def hand_with_aka_tiles(s):
    """hand.rs:14-71 + tile37_to_vec: tenhou-format string -> list of mjai tile names (akas last)."""
    names = [f"{n}{c}" for c in "mps" for n in range(1, 10)] + ["E", "S", "W", "N", "P", "F", "C", "5mr", "5pr", "5sr"]
    cnt = [0] * 37
    stack = []
    for ch in s:
        if ch.isdigit():
            stack.append(int(ch))
        elif ch in "mpsz":
            for t in stack:
                if t == 0:
                    cnt[{"m": 34, "p": 35, "s": 36}[ch]] += 1
                else:
                    cnt["mpsz".index(ch) * 9 + t - 1] += 1
            stack = []
    out = []
    for tid, c in enumerate(cnt):
        out += [names[tid]] * (c if tid < 34 else min(c, 1))
    return out


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def conv_value(v):
    v = v.strip()
    m = re.fullmatch(TILE_RE, v)
    if m:
        return m.group(1).strip()
    if v in ("true", "false"):
        return v == "true"
    if re.fullmatch(r"-?\d+", v):
        return int(v)
    m = re.fullmatch(r"\[(-?\d+); (\d+)\]", v)
    if m:
        return [int(m.group(1))] * int(m.group(2))
    m = re.fullmatch(r"\[" + TILE_RE + r"; (\d+)\]", v)
    if m:
        return [m.group(1).strip()] * int(m.group(2))
    m = re.fullmatch(r"t!\[(.*)\]", v, re.S)
    if m:
        return [x.strip() for x in m.group(1).split(",") if x.strip()]
    if v.startswith("["):
        return [conv_value(x) for x in split_top(v[1:-1])]
    m = re.search(r'hand_with_aka\("([^"]+)"\)', v)
    if m:
        return hand_with_aka_tiles(m.group(1))
    raise ValueError(v)


KIND = {"StartKyoku": "start_kyoku", "Tsumo": "tsumo", "Dahai": "dahai", "Chi": "chi", "Pon": "pon", "Daiminkan": "daiminkan",
        "Kakan": "kakan", "Ankan": "ankan", "Dora": "dora", "Reach": "reach", "ReachAccepted": "reach_accepted"}


def conv_event(text):
    m = re.match(r"Event::(\w+)\s*\{(.*)\}\s*$", text.strip(), re.S)
    assert m, text
    ev = {"type": KIND[m.group(1)]}
    for part in split_top(m.group(2)):
        if not part:
            continue
        k, v = part.split(":", 1)
        ev[k.strip()] = conv_value(v)
    return ev


def statements(body):
    """Split a function body into top-level statements (terminated by ';' at depth 0, or a `for`/block)."""
    out, depth, cur, i, n = [], 0, "", 0, len(body)
    while i < n:
        if body.startswith('r#"', i):
            j = body.index('"#', i + 3)
            cur += body[i:j + 2]
            i = j + 2
            continue
        ch = body[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        cur += ch
        i += 1
        if depth == 0 and (ch == ";" or (ch == "}" and cur.lstrip().startswith("for "))):
            out.append(cur.strip())
            cur = ""
    return out


def main():
    src = open(SRC).read()
    tests = {}
    for m in re.finditer(r"#\[test\]\s*fn (\w+)\(\) \{", src):
        name = m.group(1)
        start = m.end()
        depth, i = 1, start
        while depth:
            if src.startswith('r#"', i):
                i = src.index('"#', i + 3) + 2
                continue
            if src[i] == "{":
                depth += 1
            elif src[i] == "}":
                depth -= 1
            i += 1
        body = src[start:i - 1]
        # drop // comments (none of the embedded JSON logs contains "//")
        body = "\n".join(l.split("//")[0] for l in body.split("\n"))
        line0 = src[:m.start()].count("\n") + 1
        steps, logs = [], {}
        for st in statements(body):
            st_flat = " ".join(st.split())
            mm = re.match(r'let (\w+) = r#"(.*)"#;$', st, re.S)
            if mm:
                logs[mm.group(1)] = [json.loads(l) for l in mm.group(2).strip().split("\n")]
                continue
            mm = re.match(r"let (?:mut )?(\w+) = PlayerState::new\((\d)\);", st_flat)
            if mm:
                steps.append({"new": int(mm.group(2)), "var": mm.group(1)})
                continue
            mm = re.match(r"let (?:mut )?(\w+) = PlayerState::from_log\((\d), (\w+)\);", st_flat)
            if mm:
                steps.append({"new": int(mm.group(2)), "var": mm.group(1)})
                for ev in logs[mm.group(3)]:
                    steps.append({"ev": ev, "on": mm.group(1)})
                continue
            mm = re.match(r'let (?:mut )?(\w+) = PlayerState \{ tehai: hand\("([^"]+)"\)\.unwrap\(\), tehai_len_div3: (\d), \.\.Default::default\(\) \};', st_flat)
            if mm:
                steps.append({"new": 0, "var": mm.group(1)})
                steps.append({"set_tehai": mm.group(2), "len_div3": int(mm.group(3)), "on": mm.group(1)})
                continue
            mm = re.match(r'(\w+)\.tehai = hand\("([^"]+)"\)\.unwrap\(\);', st_flat)
            if mm:
                steps.append({"set_tehai": mm.group(2), "len_div3": None, "on": mm.group(1)})
                continue
            mm = re.match(r"(\w+)\.update_waits_and_furiten\(\);", st_flat)
            if mm:
                steps.append({"call": "update_waits_and_furiten", "on": mm.group(1)})
                continue
            mm = re.match(r"(\w+)\.set_can_chi_from_tile\(" + TILE_RE + r"\);", st_flat)
            if mm:
                steps.append({"call": "set_can_chi_from_tile", "arg": mm.group(2), "on": mm.group(1)})
                continue
            mm = re.match(r"(?:let (\w+) = )?(\w+)\s*\.test_update\(&(Event::.*)\);$", st_flat, re.S)
            if mm:
                steps.append({"ev": conv_event(mm.group(3)), "bind": mm.group(1), "on": mm.group(2)})
                continue
            mm = re.match(r'(?:let (\w+) = )?(\w+)\s*\.test_update_json\(\s*r#"(.*)"#,?\s*\);$', st_flat, re.S)
            if mm:
                steps.append({"ev": json.loads(mm.group(3)), "bind": mm.group(1), "on": mm.group(2)})
                continue
            mm = re.match(r"let (?:mut )?(\w+) = (\w+)\.clone\(\);", st_flat)
            if mm:
                steps.append({"clone": mm.group(2), "var": mm.group(1)})
                continue
            steps.append({"assert": st_flat})
        tests[name] = {"line": line0, "steps": steps}
    with open(OUT, "w") as f:
        json.dump(tests, f, indent=0)
    pats = {}
    for t in tests.values():
        for s in t["steps"]:
            if "assert" in s:
                pats[re.sub(r"\d+|t!\([^)]*\)|tuz!\([^)]*\)", "#", s["assert"])] = s["assert"]
    print({k: len(v["steps"]) for k, v in tests.items()})
    for k in sorted(pats):
        print("  ", pats[k][:160])


if __name__ == "__main__":
    main()
