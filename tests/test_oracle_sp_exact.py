"""A second, independently shaped reading of the single-player calculator (algo/sp/calc.rs:447-637) for the oracle's sp.cc: the
draw / discard recursion written directly as probabilities over the turns, in EXACT rationals — no probability tables, no f32, no
memo keyed by levels — and compared with oracle/sp.cc's f32 results (relative 2e-5).

What it re-derives, from the calculator's own model (one tile leaves the unseen pool per turn; only draws that lower the shanten
number are kept; the discard after such a draw is the one with the larger integer EV, ties by discard priority):
  * prob(i, j) = P(no useful tile on turns i .. j-1) * count / (left - j)  instead of  tsumo_prob[count][j] * not_tsumo[j] / not_tsumo[i]
    (calc.rs:135-167, 486-503), including the table's cut-off once every remaining tile is a useful one;
  * the turn bounds: a draw on the last turn cannot be followed by another one (calc.rs:533), tenpai at 1-shanten counts the
    draw itself (calc.rs:529-531);
  * ippatsu / haitei / double-riichi han bonuses per (i, j) (calc.rs:505-521) — exercised with a closed hand that does not
    prefer riichi (menzen tsumo only) and with haitei on, so the score vector is the plain point table;
  * the per-turn choice of the discard by the truncated EV (calc.rs:606-629) and the red-five draw entries (state.rs:150-165).
The second test widens it to what `get_score` does for a hand that would declare riichi (riichi + menzen tsumo han, ippatsu and
double-riichi bonuses per (i, j), the ura-dora expectation from the wall's own counts with one indicator and from the statistics
table with two or three, calc.rs:640-758), to open hands (a pon or a chi, doras in the melds, no additional han) and to 14-tile
roots (one candidate per discard that keeps the shanten number, red five last).
The shanten numbers, agari (fu / han) and points come from the oracle functions that the reference's KATs pin
(tests/test_oracle_kats.py).  CPU only."""
from fractions import Fraction
from functools import lru_cache

import numpy as np
import pytest

DISCARD_PRIO = [6, 5, 4, 3, 2, 3, 4, 5, 6] * 3 + [7] * 7 + [1, 1, 1, 0]  # tile.rs:21-28


def _tsumo_total(o, is_oya, fu, han):
    out = np.zeros(3, dtype=np.int32)
    assert o.lib().mjo_point(int(is_oya), fu, han, o.ptr(out)) == 0
    return int(out[1]) * 3 if is_oya else int(out[1]) * 2 + int(out[2])


URADORA_PROB = [  # algo/data/uradora_prob_table.txt (data restated; row = number of indicators - 1, column = ura doras that hit)
    [0.639485, 0.327801, 0.0327134, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.406736, 0.42281, 0.147966, 0.021674, 0.0008142, 0, 0, 0, 0, 0, 0, 0, 0],
    [0.257516, 0.406819, 0.246851, 0.0757724, 0.0122266, 0.0008004, 1.43e-5, 0, 0, 0, 0, 0, 0],
    [0.162199, 0.346513, 0.301539, 0.142396, 0.0401276, 0.0066491, 0.0005575, 1.85e-5, 0, 0, 0, 0, 0],
    [0.101768, 0.275319, 0.313742, 0.20189, 0.081774, 0.0215394, 0.0035918, 0.0003607, 1.52e-5, 3e-7, 0, 0, 0]]


class ExactSP:
    """The recursion in exact rationals.  Defaults: closed hand (len_div3 = 4), no riichi preference — additional han = 1 (menzen
    tsumo), han bonus only from haitei.  With `prefer_riichi` (closed hands): riichi + menzen tsumo, ippatsu / double-riichi
    bonuses and the ura-dora expectation (calc.rs:688-746); with melds: an open hand, no additional han."""

    def __init__(self, o, wall, akas_wall, T, bakaze, jikaze, dora_ind, calc_haitei, len_div3=4, melds=None, is_menzen=True,
                 prefer_riichi=False, calc_double_riichi=False, num_doras_in_fuuro=0):
        self.o, self.T, self.bakaze, self.jikaze, self.dora_ind, self.haitei = o, T, bakaze, jikaze, dora_ind, calc_haitei
        self.n_left = int(sum(wall))
        self.root_wall, self.root_akas_wall = tuple(wall), tuple(akas_wall)
        self.ld3, self.melds, self.is_menzen = len_div3, melds or {}, is_menzen
        self.riichi = is_menzen and prefer_riichi
        self.add_han = 2 if self.riichi else 1 if is_menzen else 0
        self.double_riichi, self.fuuro_doras = calc_double_riichi, num_doras_in_fuuro

    def shanten(self, hand):
        return self.o.calc_shanten(np.array(hand, dtype=np.uint8), self.ld3)

    def entries(self, hand, wall, akas_wall, L):
        """(tile37, count) of the draws that lower the shanten number, reference order (state.rs:128-173)."""
        out = []
        for t in range(34):
            c = wall[t]
            if c == 0:
                continue
            h = list(hand)
            h[t] += 1
            if self.shanten(h) != L - 1:
                continue
            k = {4: 0, 13: 1, 22: 2}.get(t)
            if k is not None and akas_wall[k]:
                if c >= 2:
                    out.append((t, c - 1))
                out.append((34 + k, 1))
            else:
                out.append((t, c))
        return out

    def score(self, hand14, akas_hand, win_tile, wall_after):
        """get_score (calc.rs:640-758): the four scores for 0..3 extra han (exact rationals), or None without a yaku."""
        t = win_tile if win_tile < 34 else (4, 13, 22)[win_tile - 34]
        nxt = self.o.lib().mjo_tile_next
        doras = sum(hand14[nxt(d)] for d in self.dora_ind) + sum(akas_hand) + self.fuuro_doras
        r = self.o.agari(np.array(hand14, dtype=np.uint8), t, False, mode=0, additional_hans=self.add_han, doras=doras,
                         bakaze=self.bakaze, jikaze=self.jikaze, **self.melds)
        if r is None:
            return None
        is_oya = self.jikaze == 27
        if r[0] == "yakuman":
            return [Fraction(16000 * r[1] * 3 if is_oya else 8000 * r[1] * 2 + 16000 * r[1])] * 4
        fu, han = r[1], r[2]
        if self.riichi and len(self.dora_ind) == 1:  # ura dora from the wall's own counts (calc.rs:693-727)
            n_ind, n_left = [0] * 5, sum(wall_after)
            for tid, c in enumerate(hand14):
                if c:
                    prev = next(q for q in range(34) if nxt(q) == tid)
                    n_ind[c] += wall_after[prev]
            probs = [Fraction(n_left - sum(n_ind), n_left)] + [Fraction(n_ind[k], n_left) for k in range(1, 5)]
        elif self.riichi and len(self.dora_ind) > 1:  # the statistics table (calc.rs:728-744)
            probs = [Fraction(x).limit_denominator(10 ** 9) for x in URADORA_PROB[len(self.dora_ind) - 1]]
        else:
            probs = [Fraction(1)]
        return [sum(p * _tsumo_total(self.o, is_oya, fu, han + i + j) for j, p in enumerate(probs) if p) for i in range(4)]

    @lru_cache(maxsize=None)
    def draw(self, hand, akas_hand, wall, akas_wall, L):
        """values[i] = (tenpai, win, ev) of a 13-tile state at shanten L when turn i is next (draw_without_tegawari)."""
        T, n = self.T, self.n_left
        ent = self.entries(hand, wall, akas_wall, L)
        R = sum(c for _, c in ent)
        ten, win, ev = [Fraction(0)] * T, [Fraction(0)] * T, [Fraction(0)] * T
        for tile, c in ent:
            t = tile if tile < 34 else (4, 13, 22)[tile - 34]
            h = list(hand); h[t] += 1
            w = list(wall); w[t] -= 1
            ah, aw = list(akas_hand), list(akas_wall)
            if tile >= 34:
                ah[tile - 34], aw[tile - 34] = 1, 0
            if L > 0:
                nxt = self.discard(tuple(h), tuple(ah), tuple(w), tuple(aw), L - 1)
                scores = None
            else:
                scores = self.score(h, ah, tile, w)
                if scores is None:
                    continue  # no yaku with this tile
            for i in range(T):
                # a turn that cannot be reached without a useful draw (every tile left is one): the table's row is zero there and
                # the reference leaves the values at zero (calc.rs:488-497)
                if any(n - k - R <= 0 for k in range(i)):
                    break
                no = Fraction(1)  # P(no useful tile on turns i .. j-1 | none before i)
                for j in range(i, T):
                    left = n - j
                    if left <= 0 or no == 0:
                        break
                    p = no * Fraction(c, left)
                    if L == 0:
                        win[i] += p
                        han_plus = int(self.riichi and self.double_riichi and i == 0) + int(self.riichi and j == i) + int(self.haitei and j == T - 1)
                        ev[i] += p * scores[han_plus]  # calc.rs:505-521
                    else:
                        if L == 1:
                            ten[i] += p
                        if j < T - 1:
                            if L > 1:
                                ten[i] += p * nxt[0][j + 1]
                            win[i] += p * nxt[1][j + 1]
                            ev[i] += p * nxt[2][j + 1]
                    no *= Fraction(max(left - R, 0), left)
        return tuple(ten), tuple(win), tuple(ev)

    @lru_cache(maxsize=None)
    def discard(self, hand, akas_hand, wall, akas_wall, L):
        """Per turn the discard with the larger truncated EV, ties by priority (discard_slow, calc.rs:563-637)."""
        T = self.T
        best = [None] * T
        vals = [None] * T
        for d in range(34):
            if hand[d] == 0:
                continue
            h = list(hand); h[d] -= 1
            if self.shanten(h) != L:
                continue
            ah = list(akas_hand)
            tile = d
            k = {4: 0, 13: 1, 22: 2}.get(d)
            if k is not None and akas_hand[k] and hand[d] == 1:  # the red five goes last (state.rs:116-121)
                tile, ah[k] = 34 + k, 0
            v = self.draw(tuple(h), tuple(ah), wall, akas_wall, L)
            for i in range(T):
                key = (int(v[2][i]), DISCARD_PRIO[tile], -tile)  # value, then cmp_discard_priority (tile.rs:169-177)
                if best[i] is None or key > best[i]:
                    best[i], vals[i] = key, (v[0][i], v[1][i], v[2][i])
        return tuple(x[0] for x in vals), tuple(x[1] for x in vals), tuple(x[2] for x in vals)


def _cases():
    """13-tile hands near completion: four random sets + a pair, one tile removed, then up to two tiles swapped for random ones."""
    rng = np.random.default_rng(20260924)
    out = []
    while len(out) < 24:
        cnt = np.zeros(34, dtype=np.int64)
        for _ in range(4):
            if rng.random() < 0.65:
                s_, p_ = int(rng.integers(0, 3)), int(rng.integers(0, 7))
                cnt[9 * s_ + p_:9 * s_ + p_ + 3] += 1
            else:
                cnt[int(rng.integers(0, 34))] += 3
        cnt[int(rng.integers(0, 34))] += 2
        if cnt.max() > 4:
            continue
        cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1
        for _ in range(int(rng.integers(0, 3))):
            cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1
            t = int(rng.integers(0, 34))
            while cnt[t] >= 4:
                t = int(rng.integers(0, 34))
            cnt[t] += 1
        rest = np.repeat(np.arange(34), 4 - cnt)
        rng.shuffle(rest)
        seen_extra = np.bincount(rest[:int(rng.integers(15, 75))], minlength=34)
        out.append((cnt, seen_extra, int(rng.integers(2, 8)), 27 + int(rng.integers(0, 2)), 27 + int(rng.integers(0, 4)),
                    int(rng.integers(0, 34)), bool(rng.integers(0, 2)), tuple(int(x) for x in rng.integers(0, 2, 3))))
    return out


@pytest.mark.parametrize("case", range(24))
def test_oracle_sp_against_exact_rational_recursion(oracle, case):
    hand, seen_extra, T, bakaze, jikaze, dora, haitei, akas_seen_bits = _cases()[case]
    L = oracle.calc_shanten(hand.astype(np.uint8), 4)
    if not (0 <= L <= 2) or T < L:
        pytest.skip("outside the recursion's range for this test")
    seen = hand + seen_extra  # seen_extra is drawn from the tiles the hand does not hold: never more than four of a kind
    if seen[dora] == hand[dora] and seen[dora] < 4:
        seen[dora] += 1  # the indicator itself is a seen tile
    wall = [4 - int(x) for x in seen]
    # red fives: in the hand if the hand holds the five and the bit says so; otherwise seen elsewhere or still in the wall
    akas_hand = tuple(int(akas_seen_bits[k] and hand[t] > 0) for k, t in enumerate((4, 13, 22)))
    akas_seen = tuple(int(akas_hand[k] or (akas_seen_bits[k] and seen[t] > hand[t]) or wall[t] == 0) for k, t in enumerate((4, 13, 22)))
    akas_wall = tuple(1 - a for a in akas_seen)
    got = oracle.sp_calc(hand.astype(np.uint8), seen.astype(np.uint8), jikaze=jikaze, bakaze=bakaze, tsumos_left=T, cur_shanten=L,
                         can_discard=False, prefer_riichi=False, calc_haitei=haitei, dora_indicators=[dora], akas_in_hand=akas_hand,
                         akas_seen=akas_seen, sort_result=False)
    assert len(got) == 1
    X = ExactSP(oracle, wall, akas_wall, T, bakaze, jikaze, [dora], haitei)
    ten, win, ev = X.draw(tuple(int(x) for x in hand), akas_hand, tuple(wall), akas_wall, L)
    c = got[0]
    n = len(c["win_probs"])
    assert n == T
    for i in range(T):
        want_t = 1.0 if L == 0 else float(ten[i])
        assert abs(float(c["win_probs"][i]) - float(win[i])) <= 2e-5 * max(1.0, float(win[i])), (i, c["win_probs"][i], float(win[i]))
        assert abs(float(c["exp_values"][i]) - float(ev[i])) <= 2e-5 * max(1.0, float(ev[i])), (i, c["exp_values"][i], float(ev[i]))
        assert abs(min(max(float(c["tenpai_probs"][i]), 0.0), 1.0) - min(want_t, 1.0)) <= 2e-5, (i, c["tenpai_probs"][i], want_t)


def _hand_near_completion(rng, n_sets):
    """`n_sets` random sets + a pair, then one or two tiles swapped for random ones (a 3*n_sets + 2 tile hand at 0..2 shanten)."""
    while True:
        cnt = np.zeros(34, dtype=np.int64)
        for _ in range(n_sets):
            if rng.random() < 0.7:
                s_, p_ = int(rng.integers(0, 3)), int(rng.integers(0, 7))
                cnt[9 * s_ + p_:9 * s_ + p_ + 3] += 1
            else:
                cnt[int(rng.integers(0, 34))] += 3
        cnt[int(rng.integers(0, 34))] += 2
        for _ in range(int(rng.integers(1, 3))):
            cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1
            cnt[int(rng.integers(0, 34))] += 1
        if cnt.max() <= 4:
            return cnt


def _cases_wide():
    """Riichi scoring (one indicator: exact ura; two or three: the table; double riichi), open hands with a pon / chi, and
    14-tile roots (one candidate per keeping discard)."""
    rng = np.random.default_rng(7741)
    out = []
    while len(out) < 36:
        kind = ("riichi1", "riichiN", "open", "root14")[len(out) % 4]
        melds, n_sets, fuuro_doras = {}, 4, 0
        if kind == "open":
            n_sets = 3
            if rng.random() < 0.5:
                melds = dict(pons=[int(rng.integers(27, 34))])  # an honour pon: often the yaku of the hand
            else:
                melds = dict(chis=[9 * int(rng.integers(0, 3)) + int(rng.integers(0, 7))])
            fuuro_doras = int(rng.integers(0, 2))
        cnt = _hand_near_completion(rng, n_sets)
        meld_cnt = np.zeros(34, dtype=np.int64)
        for t in melds.get("pons", []):
            meld_cnt[t] += 3
        for t in melds.get("chis", []):
            meld_cnt[t:t + 3] += 1
        if (cnt + meld_cnt).max() > 4:
            continue
        if kind != "root14":
            cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1  # a 3n+1 hand
        n_ind = 1 if kind in ("riichi1", "open") else int(rng.integers(1, 4))
        inds = [int(x) for x in rng.integers(0, 34, n_ind)]
        vis = cnt + meld_cnt
        for d in inds:
            vis[d] += 1
        if vis.max() > 4:
            continue
        rest = np.repeat(np.arange(34), 4 - vis)
        rng.shuffle(rest)
        seen = vis + np.bincount(rest[:int(rng.integers(10, 70))], minlength=34)
        out.append(dict(kind=kind, hand=cnt, seen=seen, melds=melds, fuuro_doras=fuuro_doras, inds=inds, T=int(rng.integers(2, 7)),
                        bakaze=27 + int(rng.integers(0, 2)), jikaze=27 + int(rng.integers(0, 4)), haitei=bool(rng.integers(0, 2)),
                        double_riichi=bool(rng.integers(0, 2)), akas=tuple(int(x) for x in rng.integers(0, 2, 3))))
    return out


@pytest.mark.parametrize("case", range(36))
def test_oracle_sp_riichi_open_hands_and_roots_against_exact_recursion(oracle, case):
    c = _cases_wide()[case]
    hand, seen, T, kind = c["hand"], c["seen"], c["T"], c["kind"]
    ld3 = 3 if kind == "open" else 4
    is_menzen = kind != "open"
    root14 = kind == "root14"
    L = oracle.calc_shanten(hand.astype(np.uint8), ld3)
    if root14 and L == -1:
        pytest.skip("complete hand")
    L = max(L, 0)
    if not (0 <= L <= 2) or T < L:
        pytest.skip("outside the recursion's range for this test")
    wall = [4 - int(x) for x in seen]
    akas_hand = tuple(int(c["akas"][k] and hand[t] > 0) for k, t in enumerate((4, 13, 22)))
    akas_seen = tuple(int(akas_hand[k] or (c["akas"][k] and seen[t] > hand[t]) or wall[t] == 0) for k, t in enumerate((4, 13, 22)))
    akas_wall = tuple(1 - a for a in akas_seen)
    riichi = kind in ("riichi1", "riichiN", "root14")
    got = oracle.sp_calc(hand.astype(np.uint8), seen.astype(np.uint8), len_div3=ld3, is_menzen=is_menzen, jikaze=c["jikaze"],
                         bakaze=c["bakaze"], tsumos_left=T, cur_shanten=L, can_discard=root14, prefer_riichi=riichi,
                         calc_double_riichi=c["double_riichi"] and riichi, calc_haitei=c["haitei"], dora_indicators=c["inds"],
                         akas_in_hand=akas_hand, akas_seen=akas_seen, num_doras_in_fuuro=c["fuuro_doras"], sort_result=False,
                         **c["melds"])
    X = ExactSP(oracle, wall, akas_wall, T, c["bakaze"], c["jikaze"], c["inds"], c["haitei"], len_div3=ld3, melds=c["melds"],
                is_menzen=is_menzen, prefer_riichi=riichi, calc_double_riichi=c["double_riichi"] and riichi,
                num_doras_in_fuuro=c["fuuro_doras"])
    h0 = tuple(int(x) for x in hand)
    if root14:
        # one candidate per discard that keeps the shanten number (calc.rs:205-247); the red five leaves last (state.rs:116-121)
        want = {}
        for d in range(34):
            if h0[d] == 0:
                continue
            h = list(h0); h[d] -= 1
            if X.shanten(h) != L:
                continue
            ah = list(akas_hand)
            tile = d
            k = {4: 0, 13: 1, 22: 2}.get(d)
            if k is not None and akas_hand[k] and h0[d] == 1:
                tile, ah[k] = 34 + k, 0
            want[tile] = X.draw(tuple(h), tuple(ah), tuple(wall), akas_wall, L)
        assert sorted(x["tile"] for x in got) == sorted(want)
        pairs = [(x, want[x["tile"]]) for x in got]
        # the sorted form (calc.rs:181-188, Candidate::cmp candidate.rs:73-106): same candidates, turn-0 EV descending, then win
        # probability, then tenpai probability — checked wherever the exact values are clearly apart
        srt = oracle.sp_calc(hand.astype(np.uint8), seen.astype(np.uint8), len_div3=ld3, is_menzen=is_menzen, jikaze=c["jikaze"],
                             bakaze=c["bakaze"], tsumos_left=T, cur_shanten=L, can_discard=True, prefer_riichi=riichi,
                             calc_double_riichi=c["double_riichi"] and riichi, calc_haitei=c["haitei"], dora_indicators=c["inds"],
                             akas_in_hand=akas_hand, akas_seen=akas_seen, num_doras_in_fuuro=c["fuuro_doras"], sort_result=True,
                             **c["melds"])
        assert sorted(x["tile"] for x in srt) == sorted(want)
        key = lambda t: (float(want[t][2][0]), float(want[t][1][0]), 1.0 if L == 0 else float(want[t][0][0]))
        for a_, b_ in zip(srt, srt[1:]):
            ka, kb = key(a_["tile"]), key(b_["tile"])
            for x, y in zip(ka, kb):
                if abs(x - y) > 1e-4 * max(1.0, abs(x)):
                    assert x > y, (a_["tile"], b_["tile"], ka, kb)
                    break
    else:
        assert len(got) == 1
        pairs = [(got[0], X.draw(h0, akas_hand, tuple(wall), akas_wall, L))]
    for cand, (ten, win, ev) in pairs:
        assert len(cand["win_probs"]) == T
        for i in range(T):
            want_t = 1.0 if L == 0 else float(ten[i])
            assert abs(float(cand["win_probs"][i]) - float(win[i])) <= 2e-5 * max(1.0, float(win[i])), (i, cand["win_probs"][i], float(win[i]))
            assert abs(float(cand["exp_values"][i]) - float(ev[i])) <= 3e-5 * max(1.0, float(ev[i])), (i, cand["exp_values"][i], float(ev[i]))
            assert abs(min(max(float(cand["tenpai_probs"][i]), 0.0), 1.0) - min(want_t, 1.0)) <= 2e-5, (i, cand["tenpai_probs"][i], want_t)


def _cases_deep():
    """Closed 13-tile hands at exactly 3 shanten (the deepest graphs the encoder builds), 3-5 draws left."""
    rng = np.random.default_rng(31337)
    out = []
    while len(out) < 60:
        cnt = _hand_near_completion(rng, 4)
        for _ in range(3):
            cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1
            t = int(rng.integers(0, 34))
            while cnt[t] >= 4:
                t = int(rng.integers(0, 34))
            cnt[t] += 1
        cnt[int(rng.choice(np.flatnonzero(cnt)))] -= 1
        if cnt.max() > 4:
            continue
        out.append((cnt, int(rng.integers(3, 6)), 27 + int(rng.integers(0, 2)), 27 + int(rng.integers(0, 4)), int(rng.integers(0, 34)),
                    int(rng.integers(20, 60)), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("case", range(6))
def test_oracle_sp_three_shanten_against_exact_recursion(oracle, case):
    deep = [c for c in _cases_deep() if oracle.calc_shanten(c[0].astype(np.uint8), 4) == 3]
    assert len(deep) >= 6
    hand, T, bakaze, jikaze, dora, n_seen, sub = deep[case]
    rng = np.random.default_rng(sub)
    vis = hand.copy()
    if vis[dora] < 4:
        vis[dora] += 1
    rest = np.repeat(np.arange(34), 4 - vis)
    rng.shuffle(rest)
    seen = vis + np.bincount(rest[:n_seen], minlength=34)
    wall = [4 - int(x) for x in seen]
    akas_seen = tuple(int(wall[t] == 0) for t in (4, 13, 22))
    akas_wall = tuple(1 - a for a in akas_seen)
    got = oracle.sp_calc(hand.astype(np.uint8), seen.astype(np.uint8), jikaze=jikaze, bakaze=bakaze, tsumos_left=T, cur_shanten=3,
                         can_discard=False, prefer_riichi=True, calc_haitei=True, dora_indicators=[dora], akas_in_hand=(0, 0, 0),
                         akas_seen=akas_seen, sort_result=False)
    assert len(got) == 1
    X = ExactSP(oracle, wall, akas_wall, T, bakaze, jikaze, [dora], True, prefer_riichi=True)
    ten, win, ev = X.draw(tuple(int(x) for x in hand), (0, 0, 0), tuple(wall), akas_wall, 3)
    c = got[0]
    for i in range(T):
        assert abs(float(c["win_probs"][i]) - float(win[i])) <= 3e-5 * max(1.0, float(win[i])) + 1e-9, (i, c["win_probs"][i], float(win[i]))
        assert abs(float(c["exp_values"][i]) - float(ev[i])) <= 3e-5 * max(1.0, float(ev[i])), (i, c["exp_values"][i], float(ev[i]))
        assert abs(float(c["tenpai_probs"][i]) - float(ten[i])) <= 3e-5, (i, c["tenpai_probs"][i], float(ten[i]))
