"""A second reading of the PRIVATE discard rows of the obs (the rows no log-driven encoder can see): keep-shanten / next-shanten
discards (state/update.rs:881-912) and the unconditional-tenpai discards (state/agent_helper.rs:88-197).

oracle/state.cc maintains these incrementally like the reference (shanten number adjusted by one on a discard / kan, candidate
sets recomputed at 3n+2).  Here they are recomputed FROM SCRATCH for every discard decision of tenpai-seeking games, by brute
force over the tiles with nothing but `calc_shanten` / `agari(has_yaku)` (KAT-pinned) and facts replayed from the mjai event log
(the seat's own discards for furiten, its melds for the yaku check, riichi acceptance) — a different program shape, so a slip in
the incremental bookkeeping of a rare path (kakan / ankan shanten shortcuts, forbidden tiles after a call, furiten by an old
discard) cannot hide in both.  The device encoder equals oracle/obs.cc bit for bit, so this pins both.  CPU only."""
import numpy as np

import parity_util

DEAKA = {34: 4, 35: 13, 36: 22}


TILE_ID = {}


def _d(t):
    if isinstance(t, str):
        t = TILE_ID[t]
    return DEAKA.get(int(t), int(t))


class SeatFacts:
    """What the unconditional-tenpai test needs beyond the snapshot, replayed from the event log."""

    def __init__(self):
        self.discarded = np.zeros(34, dtype=bool)
        self.chis, self.pons, self.minkans, self.ankans = [], [], [], []
        self.riichi_accepted = False
        self.sets = None  # (keep, next) of the last decision before riichi was accepted

    def reset(self):
        self.__init__()


def _replay(facts, events, upto):
    """Apply events[done:upto] to the four seats' facts."""
    for ev in events[facts["done"]:upto]:
        t = ev["type"]
        if t == "start_kyoku":
            for s in range(4):
                facts[s].reset()
        elif t == "dahai":
            facts[ev["actor"]].discarded[_d(ev["pai"])] = True
        elif t == "chi":
            facts[ev["actor"]].chis.append(min(_d(ev["pai"]), *(_d(c) for c in ev["consumed"])))
        elif t == "pon":
            facts[ev["actor"]].pons.append(_d(ev["pai"]))
        elif t == "daiminkan":
            facts[ev["actor"]].minkans.append(_d(ev["pai"]))
        elif t == "kakan":
            f = facts[ev["actor"]]
            f.pons.remove(_d(ev["pai"]))
            f.minkans.append(_d(ev["pai"]))
        elif t == "ankan":
            facts[ev["actor"]].ankans.append(_d(ev["consumed"][0]))
        elif t == "reach_accepted":
            facts[ev["actor"]].riichi_accepted = True
    facts["done"] = upto


def _unconditional_tenpai(o, sn, f, bakaze, jikaze):
    """agent_helper.rs:100-197 (34-tile form), from scratch."""
    ret = np.zeros(34, dtype=bool)
    tehai = sn["tehai"].astype(np.uint8)
    ld3 = sn["tehai_len_div3"]
    if sn["tiles_left"] == 0 or sn["shanten"] > 1 or (sn["shanten"] == 1 and not sn["has_next_shanten_discard"]):
        return ret
    lst = sn["last_self_tsumo"]
    if 0 <= lst < 37:
        if sn["waits"][_d(lst)]:
            return ret
        if f.riichi_accepted:
            if not sn["at_furiten"]:
                ret[_d(lst)] = True
            return ret
    elif o.calc_shanten(tehai, ld3) == -1:
        return ret
    cand = sn["next_shanten_discards"] if sn["shanten"] == 1 else sn["keep_shanten_discards"]
    for d in range(34):
        if not cand[d] or sn["forbidden_tiles"][d]:
            continue
        h13 = tehai.copy()
        h13[d] -= 1
        for t in range(34):
            if t == d or h13[t] == 4:
                continue
            h14 = h13.copy()
            h14[t] += 1
            if o.calc_shanten(h14, ld3) > -1:
                continue
            if f.discarded[t]:  # furiten
                ret[d] = False
                break
            if sn["tiles_seen"][t] == 4 or ret[d]:
                continue
            ret[d] = o.agari(h14, t, True, mode=2, chis=f.chis, pons=f.pons, minkans=f.minkans, ankans=f.ankans, bakaze=bakaze, jikaze=jikaze)
    return ret


def test_private_discard_rows_from_scratch(oracle):
    version = 3
    TILE_ID.update(oracle.TILE_ID)
    d0 = parity_util.DISCARD_ROW[version]
    n_games = 6
    arena = oracle.Arena(parity_util.default_seeds(n_games, 2468), deal_algo=1, enable_quick_eval=False, version=version, keep_log=True)
    facts = [dict({s: SeatFacts() for s in range(4)}, done=0) for _ in range(n_games)]
    checked = ut_rows = ut_hits = after_call = riichi_cases = furiten_blocks = 0
    for cycle in range(1400):
        rows = arena.poll()
        n = len(rows)
        if n == 0 and arena.n_live == 0:
            break
        obs, masks = arena.encode(0, n, want_obs=True)
        for r in range(n):
            g, seat, kan = (int(x) for x in rows[r])
            if kan:
                continue
            sn = arena.player_state(g, seat).snapshot()
            if not sn["cans"]["can_discard"]:
                continue
            events = arena.log(g)
            _replay(facts[g], events, len(events))
            f = facts[g][seat]
            tehai, ld3, sh = sn["tehai"].astype(np.uint8), sn["tehai_len_div3"], sn["shanten"]
            keep, nxt = np.zeros(34, dtype=bool), np.zeros(34, dtype=bool)
            for t in range(34):
                if tehai[t] == 0:
                    continue
                h = tehai.copy()
                h[t] -= 1
                after = oracle.calc_shanten(h, ld3)
                nxt[t] = after < sh
                keep[t] = after == sh
            if f.riichi_accepted:
                # update.rs:239-241: the sets are not refreshed once riichi is accepted — they stay those of the declaring turn
                keep, nxt = f.sets
            else:
                f.sets = (keep, nxt)
            assert (keep == sn["keep_shanten_discards"]).all() and (nxt == sn["next_shanten_discards"]).all(), (g, seat, cycle)
            assert bool(nxt.any()) == bool(sn["has_next_shanten_discard"])
            x = obs[r]
            assert (x[d0 + 1] == keep).all() and (x[d0 + 2] == nxt).all()
            # the stored number is the hand's own: 3n+2 minus the best discard, or the 3n+1 hand before the draw
            best = min(oracle.calc_shanten(np.where(np.arange(34) == t, tehai - 1, tehai).astype(np.uint8), ld3) for t in range(34) if tehai[t])
            if not f.riichi_accepted:
                assert max(best, 0) == sh - int(bool(nxt.any())), (best, sh)
            else:
                assert best == 0 and sh == 0
            kyoku = next(e for e in reversed(events) if e["type"] == "start_kyoku")
            bakaze = oracle.TILE_ID[kyoku["bakaze"]] if isinstance(kyoku["bakaze"], str) else int(kyoku["bakaze"])
            jikaze = 27 + (seat - (kyoku["kyoku"] - 1)) % 4
            want = _unconditional_tenpai(oracle, sn, f, bakaze, jikaze) if sh <= 1 else np.zeros(34, dtype=bool)
            assert (x[d0 + 3] == want).all(), (g, seat, cycle, np.flatnonzero(x[d0 + 3]), np.flatnonzero(want))
            checked += 1
            ut_rows += sh <= 1
            ut_hits += int(want.any())
            after_call += not (0 <= sn["last_self_tsumo"] < 37)
            riichi_cases += f.riichi_accepted
            furiten_blocks += int(sh <= 1 and f.discarded.any())
        d = parity_util.DISCARD_ROW[version]
        act = parity_util.greedy_actions(masks, rows, cycle, obs[:, d:d + 3], 0x9E3779B97F4A7C15) if n else np.zeros(0, np.int32)
        arena.commit(act)
    assert checked > 1200 and ut_rows > 150 and ut_hits > 40 and after_call > 20 and riichi_cases > 20 and furiten_blocks > 50, \
        (checked, ut_rows, ut_hits, after_call, riichi_cases, furiten_blocks)
