"""A SECOND, independently written encoder of the public-information planes of the obs tensor, driven by the mjai EVENT LOG.

oracle/obs.cc encodes from oracle/state.cc's PlayerState; both are line-by-line restatements of the reference, and the obs values
have no reference vectors.  Here the planes that are functions of public information — round / wind / honba /
kyotaku rows, dora indicators, the four discard rivers with their call / kan / tedashi / riichi annotations and `None`
paddings, the decay rows, kawa / fuuro / ankan overviews, tiles seen, last tedashi / riichi tiles, riichi flags, doras owned /
unseen — are re-derived by a small tracker that replays the game's mjai log (whose format IS pinned: golden log, serde event
lines) and never looks at the oracle's PlayerState, except for the seat's own concealed hand.  Written from
state/obs_repr.rs:126-440 and state/update.rs:125-215,311-340,425-460,496-660,692-725,780-830,955-960 in a different shape
(event replay -> per-block expected planes at Appendix C offsets) so that a misreading in one restatement shows up as a
difference.  The device encoder equals oracle/obs.cc bit for bit (-m gpu / emulator), so this anchors all three.
"""
import math

import numpy as np
import pytest

import parity_util
from oracle_lib import TILE_ID
from test_oracle_obs_layout import TOTAL, offsets

AKA0 = 34  # 5mr, 5pr, 5sr = 34, 35, 36


def deaka(t):
    return {34: 4, 35: 13, 36: 22}.get(t, t)


def dora_of(marker):  # tile.rs next(): suits wrap 9 -> 1, winds E S W N, dragons P F C
    t = deaka(marker)
    if t < 27:
        return t // 9 * 9 + (t % 9 + 1) % 9
    if t < 31:
        return 27 + (t - 27 + 1) % 4
    return 31 + (t - 31 + 1) % 3


class Viewer:
    """What seat `me` has seen of the current kyoku, from the mjai events alone (indices are RELATIVE seats)."""

    def __init__(self, me):
        self.me = me

    def rel(self, actor):
        return (actor + 4 - self.me) % 4

    def feed(self, ev):
        t = ev["type"]
        if t == "start_kyoku":
            self.bakaze = TILE_ID[ev["bakaze"]]
            self.kyoku = ev["kyoku"] - 1
            self.honba, self.kyotaku = ev["honba"], ev["kyotaku"]
            self.oya = self.rel(ev["oya"])
            self.jikaze = 27 + (4 - self.oya) % 4
            self.tiles_left = 70
            self.indicators = [TILE_ID[ev["dora_marker"]]]
            self.kawa = [[None] if r < self.oya else [] for r in range(4)]  # pad_kawa_at_start
            self.overview = [[] for _ in range(4)]
            self.fuuro = [[] for _ in range(4)]
            self.ankan = [[] for _ in range(4)]
            self.pending_kan, self.pending_cp = [], None
            self.declared, self.accepted = [False] * 4, [False] * 4
            self.last_tedashi, self.riichi_tile = [None] * 4, [None] * 4
            self.called = []  # consumed tiles of every call (they left a concealed hand in plain sight)
        elif t == "tsumo":
            self.tiles_left -= 1
        elif t == "dahai":
            r, pai = self.rel(ev["actor"]), TILE_ID[ev["pai"]]
            su = dict(tile=pai, dora=self.factor(deaka(pai)) > 0, tedashi=not ev["tsumogiri"], riichi=self.declared[r] and not self.accepted[r])
            self.kawa[r].append(dict(su=su, kan=self.pending_kan, cp=self.pending_cp))
            self.pending_kan, self.pending_cp = [], None
            self.overview[r].append(pai)
            if su["tedashi"]:
                self.last_tedashi[r] = su
            if su["riichi"]:
                self.riichi_tile[r] = su
        elif t in ("chi", "pon", "daiminkan"):
            r, pai = self.rel(ev["actor"]), TILE_ID[ev["pai"]]
            cons = [TILE_ID[c] for c in ev["consumed"]]
            self.fuuro[r].append(cons + [pai])
            self.called += cons
            if t == "daiminkan":
                self.pending_kan = self.pending_kan + [pai]
            else:
                self.pending_cp = cons
            if t != "chi":  # seats skipped by the call get an empty turn
                i = (ev["target"] + 1) % 4
                while i != ev["actor"]:
                    self.kawa[self.rel(i)].append(None)
                    i = (i + 1) % 4
        elif t == "kakan":
            r, pai = self.rel(ev["actor"]), TILE_ID[ev["pai"]]
            for f in self.fuuro[r]:
                if deaka(f[0]) == deaka(pai):
                    f.append(pai)
                    break
            self.pending_kan = self.pending_kan + [pai]
            self.called.append(pai)
        elif t == "ankan":
            r = self.rel(ev["actor"])
            cons = [TILE_ID[c] for c in ev["consumed"]]
            self.ankan[r].append(deaka(cons[0]))
            self.pending_kan = self.pending_kan + [deaka(cons[0])]
            self.called += cons
        elif t == "dora":
            self.indicators.append(TILE_ID[ev["dora_marker"]])
        elif t == "reach":
            self.declared[self.rel(ev["actor"])] = True
        elif t == "reach_accepted":
            self.accepted[self.rel(ev["actor"])] = True
            self.kyotaku += 1

    def factor(self, tid):
        return sum(1 for m in self.indicators if dora_of(m) == tid)


def tile_set(x, row, tiles):  # encode_tile_set: 4 count rows + 3 aka rows
    cnt = [0] * 34
    for t in tiles:
        d = deaka(t)
        x[row + cnt[d], d] = 1.0
        cnt[d] += 1
        if t >= AKA0:
            x[row + 4 + t - AKA0, :] = 1.0


def expected_public_planes(v, o, version, hand34, akas_in_hand):
    """Block name -> (first row, expected planes) for viewer `v`."""
    f32 = np.float32
    out = {}

    def block(name, rows=None):
        r0, w = o[name]
        a = np.zeros((w if rows is None else rows, 34), dtype=np.float32)
        out[name] = (r0, a)
        return a

    a = block("kyoku")
    a[v.kyoku, :] = 1.0
    if version == 4:
        a = block("honba_kyotaku")
        a[0, :] = f32(min(v.honba, 10)) / f32(10)
        a[1, :] = f32(min(v.kyotaku, 10)) / f32(10)
    a = block("bakaze_jikaze")
    a[0, v.bakaze] = 1.0
    a[1, v.jikaze] = 1.0
    a = block("kyoku_in_game")
    a[0, :] = f32(min(min(v.bakaze - 27, 1) * 4 + v.kyoku, 7)) / f32(7)
    tile_set(block("dora_indicators"), 0, v.indicators)

    longest = max(len(k) for k in v.kawa)
    decay = lambda turn: f32(math.exp(f32(-0.2) * f32(longest - 1 - turn)))  # noqa: E731  (the f32 exp is compared with a tolerance)

    def own_item(a, r, it):  # 4 rows: kan tiles, tile, aka, dora
        if it is None:
            return
        for k in it["kan"]:
            a[r, deaka(k)] = 1.0
        a[r + 1, deaka(it["su"]["tile"])] = 1.0
        if it["su"]["tile"] >= AKA0:
            a[r + 2, :] = 1.0
        if it["su"]["dora"]:
            a[r + 3, :] = 1.0

    def opp_item(a, r, it):  # 8 rows: chi/pon low, high, kan tiles, tile, aka, dora, tedashi, riichi
        if it is None:
            return
        if it["cp"] is not None:
            lo, hi = sorted(deaka(c) for c in it["cp"])
            a[r, lo] = 1.0
            a[r + 1, hi] = 1.0
        for k in it["kan"]:
            a[r + 2, deaka(k)] = 1.0
        su = it["su"]
        a[r + 3, deaka(su["tile"])] = 1.0
        for j, flag in enumerate((su["tile"] >= AKA0, su["dora"], su["tedashi"], su["riichi"])):
            if flag:
                a[r + 4 + j, :] = 1.0

    a = block("own_kawa")
    mine = v.kawa[0]
    for i, it in enumerate(mine[:6]):
        own_item(a, 4 * i, it)
    for i, it in enumerate(mine[::-1][:18]):
        own_item(a, 24 + 4 * i, it)
    a = block("own_kawa_decay")
    for turn, it in enumerate(mine):
        if it is not None:
            a[0, deaka(it["su"]["tile"])] = decay(turn)
    r0 = o["opp_kawa"][0]
    a = np.zeros((3 * 195, 34), dtype=np.float32)  # per opponent: 24 items x 8 rows, then the 3 decay rows
    out["opp_kawa+extra"] = (r0, a)
    for p in range(1, 4):
        base, kw = (p - 1) * 195, v.kawa[p]
        for i, it in enumerate(kw[:6]):
            opp_item(a, base + 8 * i, it)
        for i, it in enumerate(kw[::-1][:18]):
            opp_item(a, base + 48 + 8 * i, it)
        for turn, it in enumerate(kw):
            if it is not None:
                d, val = deaka(it["su"]["tile"]), decay(turn)
                a[base + 192, d] = val
                if it["su"]["tedashi"]:
                    a[base + 193, d] = val
                if it["su"]["riichi"]:
                    a[base + 194, d] = val

    a = block("kawa_overview")
    for p in range(4):
        tile_set(a, 7 * p, v.overview[p])
    a = block("fuuro_overview")
    for p in range(4):
        for k, meld in enumerate(v.fuuro[p]):
            r = 20 * p + 5 * k
            for t in meld:
                d = deaka(t)
                i = next(i for i in range(4) if a[r + i, d] == 0.0)
                a[r + i, d] = 1.0
                if t >= AKA0:
                    a[r + 4, :] = 1.0
    a = block("ankan_overview")
    for p in range(4):
        for t in v.ankan[p]:
            a[p, t] = 1.0

    # everything this seat has seen: indicators, every discard, every tile that left a hand in a call, its own hand
    seen = hand34.astype(np.int64).copy()
    akas_seen = list(akas_in_hand)
    for t in v.indicators + [t for p in range(4) for t in v.overview[p]] + v.called:
        seen[deaka(t)] += 1
        if t >= AKA0:
            akas_seen[t - AKA0] = True
    a = block("seen_tedashi_riichi_tiles")
    a[0] = seen.astype(np.float32) / f32(4)
    for j, src in enumerate((v.last_tedashi, v.riichi_tile)):
        for p in range(1, 4):
            su = src[p]
            if su is not None:
                r = 1 + 9 * j + 3 * (p - 1)
                a[r, deaka(su["tile"])] = 1.0
                if su["tile"] >= AKA0:
                    a[r + 1, :] = 1.0
                if su["dora"]:
                    a[r + 2, :] = 1.0
    a = block("riichi_declared_accepted")
    for p in range(1, 4):
        a[p - 1, :] = float(v.declared[p])
        a[3 + p - 1, :] = float(v.accepted[p])
    a = block("self_riichi_accepted")
    a[0, :] = float(v.accepted[0])

    if version == 4:
        factor = np.array([v.factor(t) for t in range(34)])
        owned = []
        for p in range(4):
            tiles = [t for m in v.fuuro[p] for t in m] + [t for k in v.ankan[p] for t in (k, k, k, k)]
            n = sum(int(factor[deaka(t)]) for t in tiles) + sum(t >= AKA0 for t in tiles)
            n += sum(k in (4, 13, 22) for k in v.ankan[p])  # an ankan of fives holds the red one (consumed = [akaize(t), t, t, t])
            owned.append(n)
        owned[0] += int((hand34 * factor).sum()) + int(sum(akas_in_hand))
        a = block("doras_owned")
        for p in range(4):
            a[p, :] = f32(min(owned[p], 12)) / f32(12)
        doras_seen = int((seen * factor).sum()) + int(sum(akas_seen))
        a = block("doras_unseen")
        a[0, :] = f32(min(len(v.indicators) * 4 + 3 - doras_seen, 23)) / f32(23)
    return out


@pytest.mark.parametrize("version", [3, 4])
def test_public_planes_follow_from_the_event_log(oracle, version):
    o = offsets(version)
    seeds = parity_util.default_seeds(8, 97531)
    arena = oracle.Arena(seeds, deal_algo=1, enable_quick_eval=False, version=version, keep_log=True)
    viewers, fed = {}, {}
    checked = kan_items = call_items = pad_rows = riichi_rows = 0
    for cycle in range(700):
        rows = arena.poll()
        n = len(rows)
        if n == 0 and arena.n_live == 0:
            break
        obs, masks = arena.encode(0, n, want_obs=True)
        logs = {}
        for r in range(n):
            g, seat, kan = (int(x) for x in rows[r])
            if kan:
                continue
            if g not in logs:
                logs[g] = arena.log(g)
            key = (g, seat)
            if key not in viewers:
                viewers[key], fed[key] = Viewer(seat), 0
            v = viewers[key]
            for ev in logs[g][fed[key]:]:
                v.feed(ev)
            fed[key] = len(logs[g])
            sn = arena.player_state(g, seat).snapshot()
            x = obs[r]
            assert x.shape[0] == TOTAL[version]
            want = expected_public_planes(v, o, version, np.asarray(sn["tehai"]), [bool(b) for b in sn["akas_in_hand"]])
            for name, (r0, a) in want.items():
                got = x[r0:r0 + a.shape[0]]
                if "decay" in name or name == "opp_kawa+extra":  # exp(-0.2 k): the reference's f32 libm vs Python's double exp
                    ok = np.allclose(got, a, rtol=0, atol=2e-7)
                else:
                    ok = np.array_equal(got, a)
                if not ok:
                    bad = np.argwhere(~np.isclose(got, a, rtol=0, atol=2e-7))[:5]
                    raise AssertionError(f"{name}: game {g} seat {seat} cycle {cycle}: rows/cols {bad.tolist()} "
                                         f"got {[float(got[i, j]) for i, j in bad]} want {[float(a[i, j]) for i, j in bad]}")
            kan_items += sum(bool(it and it["kan"]) for k in v.kawa for it in k)
            call_items += sum(bool(it and it["cp"]) for k in v.kawa for it in k)
            pad_rows += sum(it is None for k in v.kawa for it in k)
            riichi_rows += any(v.declared)
            checked += 1
        d = parity_util.DISCARD_ROW[version]
        act = parity_util.greedy_actions(masks, rows, cycle, obs[:, d:d + 3], 0x9E3779B97F4A7C15) if n else np.zeros(0, np.int32)
        arena.commit(act)
    # the interesting annotations all occurred
    assert checked > 2000 and kan_items > 0 and call_items > 100 and pad_rows > 100 and riichi_rows > 50
