"""A second reading of the invisible ("oracle") observation (arena/board.rs:679-782) for oracle/ and, through the lock-step
tests, for `mj_k_encode_oracle`: every decision row of four tenpai-seeking hanchan is rebuilt from nothing but

  * the 136-tile sequence of the kyoku, dealt again from the game seed (`deal`, pinned by tests/test_oracle_deal.py), cut the way
    `Board::init_from_seed` cuts it (board.rs:111-121: 4 x 13 haipai, 4 rinshan, 5 dora indicators, 5 ura, 70 yama) and consumed
    the way `BoardState` consumes it (yama / rinshan / dora indicators popped from the back, ura read forward), and
  * the mjai event log replayed into four plain tile multisets (no PlayerState): the other seats' hands, red fives, meld counts,
    every tile a seat can see.

Rows compared exactly: the three other seats' hand thermometers, red fives, shanten one-hot and shanten / 6 (brute-force
`calc_shanten` of the replayed hand), waits (`update.rs:930-945`: agari shape and fewer than four copies visible to THAT seat when it last acted), the
69 x 2 yama rows (next draw first), 4 x 2 rinshan, 5 x 2 dora indicators in reveal order, 5 x 2 ura.  The furiten row is checked
one way (a wait among the seat's own discards must set it; riichi / same-turn furiten need the pass history).  It also pins the
log itself: the haipai and every draw of the log must be the tiles the sequence predicts.  CPU only."""
import numpy as np

import parity_util

AKA = {34: 4, 35: 13, 36: 22}


def _de(t):
    return AKA.get(t, t)


class Replay:
    """Four tile multisets + what everybody can see, from the events of one kyoku."""

    def __init__(self, start, tid):
        self.tid = tid
        self.hands = [[tid[x] for x in h] for h in start["tehais"]]
        self.melds = [0, 0, 0, 0]
        self.discards = [[], [], [], []]
        self.public = np.zeros(34, dtype=np.int64)  # river + meld tiles that came out of a hand + revealed indicators
        self.public[_de(tid[start["dora_marker"]])] += 1
        self.yama_used = self.rinshan_used = 0
        self.pending_rinshan = -1  # seat whose next draw comes from the dead wall
        self.draws = []            # (from_rinshan, tile) in order
        # `waits` is refreshed by a seat's OWN events only (start of the kyoku, its discards, its kans; not after its riichi was
        # accepted: update.rs:352-356, 655-658) with the tiles it could see THEN — later discards of the others do not clear a wait
        self.riichi_accepted = [False] * 4
        self.seen_then = [self.public + self.counts(s) for s in range(4)]

    def apply(self, ev):
        t, tid = ev["type"], self.tid
        if t == "tsumo":
            a, pai = ev["actor"], tid[ev["pai"]]
            rin = self.pending_rinshan == a
            self.pending_rinshan = -1
            self.draws.append((rin, pai))
            if rin:
                self.rinshan_used += 1
            else:
                self.yama_used += 1
            self.hands[a].append(pai)
        elif t == "dahai":
            a, pai = ev["actor"], tid[ev["pai"]]
            self.hands[a].remove(pai)
            self.discards[a].append(_de(pai))
            self.public[_de(pai)] += 1
            if not self.riichi_accepted[a]:
                self.seen_then[a] = self.public + self.counts(a)
        elif t in ("chi", "pon", "daiminkan"):
            a = ev["actor"]
            for c in ev["consumed"]:
                self.hands[a].remove(tid[c])
                self.public[_de(tid[c])] += 1
            self.melds[a] += 1
            if t == "daiminkan":
                self.pending_rinshan = a
        elif t == "kakan":
            a, pai = ev["actor"], tid[ev["pai"]]
            self.hands[a].remove(pai)
            self.public[_de(pai)] += 1
            self.pending_rinshan = a
            self.seen_then[a] = self.public + self.counts(a)
        elif t == "ankan":
            a = ev["actor"]
            for c in ev["consumed"]:
                self.hands[a].remove(tid[c])
                self.public[_de(tid[c])] += 1
            self.melds[a] += 1
            self.pending_rinshan = a
            if not self.riichi_accepted[a]:
                self.seen_then[a] = self.public + self.counts(a)
        elif t == "reach_accepted":
            self.riichi_accepted[ev["actor"]] = True
        elif t == "dora":
            self.public[_de(tid[ev["dora_marker"]])] += 1

    def counts(self, seat):
        c = np.zeros(34, dtype=np.uint8)
        for x in self.hands[seat]:
            c[_de(x)] += 1
        return c


def _expected(o, rp, seq, p):
    """The 217 x 34 plane stack of perspective p (obs v2..v4 layout), furiten rows left at -1 (= not compared exactly)."""
    out = np.zeros((217, 34), dtype=np.float32)
    idx = 0
    fur = []
    for r in (1, 2, 3):
        s = (p + r) % 4
        c = rp.counts(s)
        ld3 = 4 - rp.melds[s]
        assert int(c.sum()) == 3 * ld3 + 1, "another seat holds 3n+1 tiles at a decision"
        for t in range(34):
            out[idx:idx + int(c[t]), t] = 1.0
        idx += 4
        for k in range(3):
            if 34 + k in rp.hands[s]:
                out[idx + k, :] = 1.0
        idx += 3
        sh = o.calc_shanten(c, ld3)
        out[idx + sh, :] = 1.0
        idx += 7
        out[idx, :] = np.float32(sh) / np.float32(6.0)
        idx += 1
        own_furiten = False
        if sh == 0:
            seen = rp.seen_then[s]  # everything public + its own hand, when the seat last refreshed its waits
            for t in range(34):
                if c[t] == 4:
                    continue
                h = c.copy()
                h[t] += 1
                if o.calc_shanten(h, ld3) == -1:
                    own_furiten |= t in rp.discards[s]
                    if seen[t] < 4:
                        out[idx, t] = 1.0
        idx += 1
        fur.append((idx, own_furiten))
        out[idx, :] = -1.0
        idx += 1

    def tile(i, t):
        out[i, _de(t)] = 1.0
        if t >= 34:
            out[i + 1, :] = 1.0

    left = 70 - rp.yama_used - rp.rinshan_used  # a dead-wall draw shortens the live wall by its last tile (board.rs: tiles_left)
    for i in range(left):
        tile(idx + 2 * i, int(seq[135 - rp.yama_used - i]))
    idx += 69 * 2
    for i in range(4 - rp.rinshan_used):
        tile(idx + 2 * i, int(seq[55 - rp.rinshan_used - i]))
    idx += 4 * 2
    for i in range(5):
        tile(idx + 2 * i, int(seq[60 - i]))
    idx += 10
    for i in range(5):
        tile(idx + 2 * i, int(seq[61 + i]))
    idx += 10
    assert idx == 217
    return out, fur


def test_invisible_obs_rebuilt_from_the_seed_and_the_log(oracle):
    n_games, seed0, version = 4, 86420, 3
    seeds = parity_util.default_seeds(n_games, seed0)
    arena = oracle.Arena(seeds, deal_algo=1, enable_quick_eval=True, version=version, keep_log=True)
    d = parity_util.DISCARD_ROW[version]
    replays = [None] * n_games   # (index of the kyoku's start event, Replay, events applied, sequence)
    checked = with_waits = furiten_seen = kans = reds = late = 0
    for cycle in range(4000):
        rows = arena.poll()
        n = len(rows)
        if n == 0 and arena.n_live == 0:
            break
        obs, masks = arena.encode(0, n, want_obs=True)
        inv = arena.encode_oracle(0, n, version)
        for r in range(n):
            g, p, _ = (int(x) for x in rows[r])
            events = arena.log(g)
            k0 = max(i for i, e in enumerate(events) if e["type"] == "start_kyoku")
            st = replays[g]
            if st is None or st[0] != k0:
                start = events[k0]
                bakaze = oracle.TILE_ID[start["bakaze"]] if isinstance(start["bakaze"], str) else int(start["bakaze"])
                seq = oracle.deal(seeds[g][0], seeds[g][1], (bakaze - 27) * 4 + start["kyoku"] - 1, start["honba"], 1)
                rp = Replay(start, oracle.TILE_ID)
                # the log's haipai and first dora indicator are the sequence's (board.rs:111-118, 206-211)
                for s in range(4):
                    assert [int(x) for x in seq[13 * s:13 * s + 13]] == rp.hands[s]
                assert oracle.TILE_ID[start["dora_marker"]] == int(seq[60])
                st = [k0, rp, k0 + 1, seq]
                replays[g] = st
            _, rp, done, seq = st
            for ev in events[done:]:
                rp.apply(ev)
            st[2] = len(events)
            # every draw of the log is the tile the sequence predicts
            ny = nr = 0
            for rin, tile in rp.draws:
                assert tile == int(seq[55 - nr] if rin else seq[135 - ny])
                nr, ny = nr + rin, ny + (not rin)
            want, fur = _expected(oracle, rp, seq, p)
            got = inv[r]
            cmp = want >= 0
            assert (got[cmp] == want[cmp]).all(), (g, p, cycle, np.argwhere((got != want) & cmp)[:5].tolist())
            for row, own in fur:
                assert got[row].min() == got[row].max() and got[row][0] in (0.0, 1.0)
                if own:
                    assert got[row][0] == 1.0, (g, p, cycle, "a wait among the seat's own discards is furiten")
                furiten_seen += int(got[row][0])
            checked += 1
            with_waits += int(any(want[16 * k + 15].any() for k in range(3)))
            kans += rp.rinshan_used > 0
            reds += int(any(x >= 34 for s in range(4) for x in rp.hands[s]))
            late += rp.yama_used > 50
        act = parity_util.greedy_actions(masks, rows, cycle, obs[:, d:d + 3], 0xBEEF) if n else np.zeros(0, np.int32)
        arena.commit(act)
    assert arena.n_live == 0
    assert checked > 1500 and with_waits > 200 and furiten_seen > 20 and kans > 10 and reds > 500 and late > 100, \
        (checked, with_waits, furiten_seen, kans, reds, late)
