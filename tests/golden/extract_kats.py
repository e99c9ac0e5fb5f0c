#!/usr/bin/env python3
"""Extract the reference's known-answer tests into JSON fixtures (run in the build container only).

Sources (all under /root/reference/libriichi/src):
  algo/shanten.rs:158-201   19 hands -> calc_all
  algo/agari.rs:920-957     15 ankan-after-riichi cases
  algo/agari.rs:960-1379    agari hands -> fu/han/yakuman
  tile.rs, rankings.rs      small unit KATs restated by hand in tests/

Only test *vectors* (hand strings, flags, expected numbers) are extracted, never code.
Output: tests/golden/kat_algo.json
"""
import json
import os
import re

SRC = "/root/reference/libriichi/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_algo.json")


def shanten_cases():
    text = open(f"{SRC}/algo/shanten.rs").read()
    text = text[text.index("mod test"):]
    out = []
    for m in re.finditer(r'hand\("([^"]+)"\)\.unwrap\(\);\s*assert_eq!\(calc_all\(&tehai, (\d)\), (-?\d+)\);', text):
        out.append(dict(hand=m.group(1), len_div3=int(m.group(2)), expect=int(m.group(3))))
    return out


def ankan_cases():
    text = open(f"{SRC}/algo/agari.rs").read()
    out = []
    for m in re.finditer(r'test_one\("([^"]+)", "([^"]+)", (\d), (true|false), (true|false)\);', text):
        out.append(dict(hand=m.group(1), tile=m.group(2), len_div3=int(m.group(3)), strict=m.group(4) == "true",
                        expect=m.group(5) == "true"))
    return out


def tu8_list(s):
    s = s.strip()
    if s == "&[]":
        return []
    m = re.match(r"&tu8!\[(.*)\]", s)
    assert m, s
    return [x.strip() for x in m.group(1).split(",") if x.strip()]


def agari_cases():
    lines = open(f"{SRC}/algo/agari.rs").read().split("\n")
    start = next(i for i, l in enumerate(lines) if "fn agari_calc()" in l)
    ctx = None
    out = []
    pending_agari = None
    i = start
    while i < len(lines):
        l = lines[i].strip()
        i += 1
        m = re.match(r'let tehai = hand\("([^"]+)"\)\.unwrap\(\);', l)
        if m:
            ctx = dict(hand=m.group(1), line=i)
            continue
        if ctx is None:
            continue
        m = re.match(r"(is_menzen|is_ron): (true|false),", l)
        if m:
            ctx[m.group(1)] = m.group(2) == "true"
            continue
        m = re.match(r"(chis|pons|minkans|ankans): (.*),$", l)
        if m:
            ctx[m.group(1)] = tu8_list(m.group(2))
            continue
        m = re.match(r"(bakaze|jikaze|winning_tile): tu8!\((\w+)\),", l)
        if m:
            ctx[m.group(1)] = m.group(2)
            continue
        m = re.match(r"calc\.is_ron = (true|false);", l)
        if m:
            ctx["is_ron"] = m.group(1) == "true"
            continue
        m = re.match(r"calc\.winning_tile = tu8!\((\w+)\);", l)
        if m:
            ctx["winning_tile"] = m.group(1)
            continue
        m = re.match(r"assert_eq!\(yaku, Agari::Normal \{ fu: (\d+), han: (\d+) \}\);", l)
        if m:
            out.append(dict(ctx, mode="search_yakus", expect=["normal", int(m.group(1)), int(m.group(2))]))
            continue
        m = re.match(r"assert_eq!\(yaku, Agari::Yakuman\((\d+)\)\);", l)
        if m:
            out.append(dict(ctx, mode="search_yakus", expect=["yakuman", int(m.group(1))]))
            continue
        if l == "assert_eq!(calc.search_yakus(), None);":
            out.append(dict(ctx, mode="search_yakus", expect=None))
            continue
        m = re.match(r"assert!\(matches!\(yaku, Agari::Normal \{ han: (\d+), \.\. \}\)\);", l)
        if m:
            out.append(dict(ctx, mode="search_yakus", expect=["han", int(m.group(1))]))
            continue
        m = re.match(r"let points = calc\.agari\((\d+), (\d+)\)\.unwrap\(\)\.point\((true|false)\);", l)
        if m:
            pending_agari = dict(ctx, mode="agari_point", additional_hans=int(m.group(1)), doras=int(m.group(2)),
                                 is_oya=m.group(3) == "true", expect={})
            continue
        if pending_agari is not None:
            m = re.match(r"(ron|tsumo_oya|tsumo_ko): (\d+),?", l)
            if m:
                pending_agari["expect"][m.group(1)] = int(m.group(2))
                if len(pending_agari["expect"]) == 3:
                    out.append(pending_agari)
                    pending_agari = None
            continue
        if l.startswith("fn ") and i - 1 > start:
            break
    return out


def main():
    data = dict(shanten=shanten_cases(), ankan_after_riichi=ankan_cases(), agari=agari_cases())
    print({k: len(v) for k, v in data.items()})
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1)


if __name__ == "__main__":
    main()
