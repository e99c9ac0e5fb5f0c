"""The reference's own known-answer tests applied to the HIP code (not only to the oracle): shanten (algo/shanten.rs:158-201),
ankan after riichi and agari with open melds / rare yaku / yakuman (algo/agari.rs:920-1379), the point table sweep
(algo/point.rs:121-153) — through the debug entry point `mj_algo_query` of the C-ABI, one device thread per query — plus
>= 10^5 generated complete hands with random melds, winds and ron / tsumo, device vs oracle (fu, han, yakuman count, points).

`check_*` take the C-ABI library handle, so tests/test_emu_device_code.py runs the same checks on the host emulation of the
device code (`-m "not gpu"`), at a reduced size."""
import json
import os

import numpy as np
import pytest

from mortal_amd import algo_query as AQ

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_algo.json")))


def _tid(o, names):
    return [o.TILE_ID[x] for x in names]


def check_reference_kats(oracle, lib=None):
    sh, ak, ag = KATS["shanten"], [c for c in KATS["ankan_after_riichi"] if not c["strict"]], KATS["agari"]
    q = AQ.queries(len(sh) + len(ak) + 2 * len(ag))
    i = 0
    for c in sh:
        q["tehai"][i] = oracle.hand(c["hand"])
        q["len_div3"][i] = c["len_div3"]
        q["op"][i] = AQ.OP_SHANTEN
        i += 1
    for c in ak:
        t = oracle.hand(c["hand"])
        tile = oracle.TILE_ID[c["tile"]]
        t[tile] += 1
        q["tehai"][i] = t
        q["len_div3"][i] = c["len_div3"]
        q["arg0"][i] = tile
        q["op"][i] = AQ.OP_ANKAN_AFTER_RIICHI
        i += 1
    for c in ag:
        for rep in range(2):  # search_yakus cases: also has_yaku; agari cases: one query (the second slot repeats it)
            q["tehai"][i] = oracle.hand(c["hand"])
            AQ.set_melds(q, i, _tid(oracle, c["chis"]), _tid(oracle, c["pons"]), _tid(oracle, c["minkans"]), _tid(oracle, c["ankans"]))
            q["is_menzen"][i] = int(c["is_menzen"])
            q["bakaze"][i] = oracle.TILE_ID[c["bakaze"]]
            q["jikaze"][i] = oracle.TILE_ID[c["jikaze"]]
            q["winning_tile"][i] = oracle.TILE_ID[c["winning_tile"]]
            q["is_ron"][i] = int(c["is_ron"])
            if c["mode"] == "search_yakus":
                q["op"][i] = AQ.OP_SEARCH_YAKUS if rep == 0 else AQ.OP_HAS_YAKU
            else:
                q["op"][i] = AQ.OP_AGARI
                q["additional_hans"][i] = c["additional_hans"]
                q["doras"][i] = c["doras"]
                q["arg0"][i] = int(c["is_oya"])
            i += 1
    r = AQ.run(q, lib)
    i = 0
    for c in sh:
        assert r["r0"][i] == c["expect"], ("shanten", c)
        i += 1
    for c in ak:
        assert bool(r["r0"][i]) == c["expect"], ("ankan_after_riichi", c)
        i += 1
    n_yakuman = n_open = 0
    for c in ag:
        a, b = r[i], r[i + 1]
        i += 2
        n_open += bool(c["chis"] or c["pons"] or c["minkans"])
        if c["mode"] == "search_yakus":
            e = c["expect"]
            if e is None:
                assert a["r0"] == 0 and b["r0"] == 0, c
            elif e[0] == "han":
                assert a["r0"] == 1 and a["r2"] == e[1], c
            elif e[0] == "yakuman":
                assert a["r0"] == 2 and a["r2"] == e[1] and b["r0"] == 1, c
                n_yakuman += 1
            else:
                assert (a["r0"], a["r1"], a["r2"]) == (1, e[1], e[2]) and b["r0"] == 1, (c, a)
        else:
            assert a["r0"] != 0 and dict(ron=int(a["p0"]), tsumo_ko=int(a["p1"]), tsumo_oya=int(a["p2"])) == c["expect"], (c, a)
    return dict(shanten=len(sh), ankan=len(ak), agari=len(ag), yakuman=n_yakuman, open=n_open)


def check_point_sweep(lib=None):
    """algo/point.rs:121-153: the table equals ceil100(fu * 2^(han+2) * k), capped at mangan — on the device's closed form."""
    cases = [(o, fu, han) for fu in list(range(20, 120, 10)) + [25] for han in range(1, 15) for o in (0, 1)
             if not ((fu == 20 or fu == 25) and han == 1)]
    q = AQ.queries(len(cases))
    q["op"] = AQ.OP_POINT
    q["arg0"], q["arg1"], q["arg2"] = np.array(cases, dtype=np.uint8).T
    r = AQ.run(q, lib)
    ceil100 = lambda x: (x + 99) // 100 * 100
    for (o, fu, han), got in zip(cases, r):
        base = fu * 2 ** (han + 2)
        base = 8000 if han >= 13 else 6000 if han >= 11 else 4000 if han >= 8 else 3000 if han >= 6 else 2000 if (han >= 5 or base >= 2000) else base
        exp = (ceil100(base * 6), ceil100(base * 2), 0) if o else (ceil100(base * 4), ceil100(base), ceil100(base * 2))
        assert (int(got["p0"]), int(got["p1"]), int(got["p2"])) == exp, (o, fu, han)
    return len(cases)


def random_complete_hands(n, seed):
    """Complete hands by construction: 4 sets (each concealed or a random chi / pon / minkan / ankan) + a pair, a random winning
    tile from the concealed part, random winds, ron / tsumo.  Biased towards the shapes random play almost never reaches: one-suit
    hands, terminal / honour sets, triplet-heavy hands, identical sequences, seven pairs, thirteen orphans."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        mode = int(rng.integers(0, 12))
        cnt = np.zeros(34, dtype=np.int64)   # concealed tiles
        total = np.zeros(34, dtype=np.int64)  # concealed + melds (4 per kan)
        melds = dict(chis=[], pons=[], minkans=[], ankans=[])
        if mode == 10:  # seven pairs
            ks = rng.choice(34, 7, replace=False)
            cnt[ks] = 2
            total = cnt.copy()
        elif mode == 11:  # thirteen orphans
            yao = [0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33]
            cnt[yao] = 1
            cnt[int(rng.choice(yao))] += 1
            total = cnt.copy()
        else:
            suits = [int(rng.integers(0, 3))] if mode in (0, 1) else [0, 1, 2]
            ok = True
            for _ in range(4):
                r = rng.random()
                p_seq = 0.15 if mode in (2, 3) else 0.6
                if r < p_seq:  # sequence
                    s = int(rng.choice(suits))
                    p = int(rng.choice([0, 6])) if mode == 4 else int(rng.integers(0, 7))
                    tiles, kind = [9 * s + p, 9 * s + p + 1, 9 * s + p + 2], "seq"
                else:
                    if mode == 4:
                        t = int(rng.choice([0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33]))
                    elif mode in (0, 1):
                        t = 9 * suits[0] + int(rng.integers(0, 9)) if rng.random() < 0.7 else int(rng.integers(27, 34))
                    else:
                        t = int(rng.integers(0, 34))
                    tiles, kind = [t] * 3, "tri"
                o = rng.random()
                if kind == "seq":
                    if o < 0.25:
                        melds["chis"].append(tiles[0])
                        for t in tiles: total[t] += 1
                    else:
                        for t in tiles: cnt[t] += 1; total[t] += 1
                else:
                    t = tiles[0]
                    if o < 0.15: melds["pons"].append(t); total[t] += 3
                    elif o < 0.25: melds["minkans"].append(t); total[t] += 4
                    elif o < 0.35: melds["ankans"].append(t); total[t] += 4
                    else: cnt[t] += 3; total[t] += 3
            if mode in (0, 1):
                pt = 9 * suits[0] + int(rng.integers(0, 9)) if rng.random() < 0.7 else int(rng.integers(27, 34))
            else:
                pt = int(rng.integers(0, 34))
            cnt[pt] += 2
            total[pt] += 2
            if not ok:
                continue
        if total.max() > 4:
            continue
        held = np.flatnonzero(cnt)
        out.append(dict(tehai=cnt.astype(np.uint8), melds=melds, winning_tile=int(rng.choice(held)), is_ron=bool(rng.integers(0, 2)),
                        bakaze=27 + int(rng.integers(0, 2)), jikaze=27 + int(rng.integers(0, 4)),
                        additional_hans=int(rng.integers(0, 3)), doras=int(rng.integers(0, 4)), is_oya=int(rng.integers(0, 2))))
    return out


def check_generated_hands(oracle, n, seed, lib=None):
    hands = random_complete_hands(n, seed)
    q = AQ.queries(2 * n)
    for k, h in enumerate(hands):
        for rep, op in enumerate((AQ.OP_SEARCH_YAKUS, AQ.OP_AGARI)):
            i = 2 * k + rep
            q["tehai"][i] = h["tehai"]
            AQ.set_melds(q, i, **h["melds"])
            q["len_div3"][i] = 4 - sum(len(v) for v in h["melds"].values())
            q["bakaze"][i], q["jikaze"][i] = h["bakaze"], h["jikaze"]
            q["winning_tile"][i] = h["winning_tile"]
            q["is_ron"][i] = int(h["is_ron"])
            q["op"][i] = op
            q["additional_hans"][i], q["doras"][i], q["arg0"][i] = h["additional_hans"], h["doras"], h["is_oya"]
    r = AQ.run(q, lib)
    stats = dict(n=n, none=0, yakuman=0, open=0, kans=0)
    for k, h in enumerate(hands):
        kw = dict(bakaze=h["bakaze"], jikaze=h["jikaze"], **h["melds"])
        want = oracle.agari(h["tehai"], h["winning_tile"], h["is_ron"], mode=1, **kw)
        a = r[2 * k]
        got = None if a["r0"] == 0 else ("yakuman", int(a["r2"])) if a["r0"] == 2 else ("normal", int(a["r1"]), int(a["r2"]))
        assert got == want, (h, got, want)
        want = oracle.agari(h["tehai"], h["winning_tile"], h["is_ron"], mode=0, additional_hans=h["additional_hans"], doras=h["doras"], **kw)
        b = r[2 * k + 1]
        got = None if b["r0"] == 0 else ("yakuman", int(b["r2"])) if b["r0"] == 2 else ("normal", int(b["r1"]), int(b["r2"]))
        assert got == want, (h, got, want)
        if want is not None:
            out = np.zeros(3, dtype=np.int32)
            if want[0] == "yakuman":
                mul = want[1]
                exp = (48000 * mul, 16000 * mul, 0) if h["is_oya"] else (32000 * mul, 8000 * mul, 16000 * mul)  # point.rs:100-112
            elif oracle.lib().mjo_point(h["is_oya"], want[1], want[2], oracle.ptr(out)) == 0:
                exp = tuple(int(x) for x in out)
            else:  # 20 / 25 fu with one han (a pinfu tsumo scored without its tsumo han): not in the reference's table (it panics)
                assert want[1] in (20, 25) and want[2] == 1, want
                exp = None
            assert exp is None or (int(b["p0"]), int(b["p1"]), int(b["p2"])) == exp, (h, want)
        stats["none"] += want is None
        stats["yakuman"] += want is not None and want[0] == "yakuman"
        stats["open"] += bool(h["melds"]["chis"] or h["melds"]["pons"] or h["melds"]["minkans"])
        stats["kans"] += bool(h["melds"]["minkans"] or h["melds"]["ankans"])
    return stats


def check_deal_divmod(lib=None, per_n=600, seed=5):
    """mj_deal.h deal_divmod (the rand-0.9.1 shuffle's `chunk % n`, `chunk / n` as multiply-high + bounded correction) == integer
    division for EVERY divisor 1..136: edge values (0, n - 1, n, multiples of n and their neighbours, 2^32 - 1) and random u32."""
    rng = np.random.default_rng(seed)
    xs, ns = [], []
    for n in range(1, 137):
        edge = [0, 1, n - 1, n, n + 1, 2 ** 32 - 1, 2 ** 32 - 2, (2 ** 32 - 1) // n * n, (2 ** 32 - 1) // n * n - 1, 2 ** 31, 2 ** 31 - 1]
        mult = rng.integers(0, 2 ** 32 // n, size=40, dtype=np.uint64) * np.uint64(n)
        x = np.concatenate([np.array(edge, dtype=np.uint64), mult, mult + np.uint64(n - 1),
                            rng.integers(0, 2 ** 32, size=per_n, dtype=np.uint64)]) & np.uint64(0xFFFFFFFF)
        xs.append(x)
        ns.append(np.full(len(x), n, dtype=np.uint64))
    x, n = np.concatenate(xs), np.concatenate(ns)
    q = AQ.queries(len(x))
    q["op"] = AQ.OP_DEAL_DIVMOD
    q["arg0"] = n.astype(np.uint8)
    q["tehai"][:, :4] = x.astype("<u4").view(np.uint8).reshape(-1, 4)
    r = AQ.run(q, lib)
    assert np.array_equal(r["r0"].astype(np.uint32).astype(np.uint64), x // n), "deal_divmod quotient"
    assert np.array_equal(r["r1"].astype(np.uint32).astype(np.uint64), x % n), "deal_divmod remainder"
    return len(x)


@pytest.mark.gpu
def test_reference_kats_on_device(oracle):
    assert check_deal_divmod() > 80_000
    st = check_reference_kats(oracle)
    assert st["shanten"] == 19 and st["agari"] == 25 and st["ankan"] >= 3 and st["open"] >= 5
    assert check_point_sweep() > 250


@pytest.mark.gpu
def test_generated_complete_hands_device_vs_oracle(oracle):
    st = check_generated_hands(oracle, 100_000, 20260924)
    assert st["open"] > 30_000 and st["kans"] > 15_000 and st["yakuman"] > 1_000 and st["none"] > 1_000, st
