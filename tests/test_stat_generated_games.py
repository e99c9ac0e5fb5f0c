"""`libriichi.stat.Stat` (stat.rs:263-441) on GENERATED games: 16 whole hanchan played by the oracle arena under the
tenpai-seeking test policy (riichi, calls, ron / tsumo, exhaustive and abortive draws, dealer repeats, busted seats all occur),
every seat's counters compared with a second reading of the same event logs that is shaped differently from stat.rs:

  * the log is cut into kyoku first and every kyoku is summarised on its own (who called, who declared / had the riichi
    accepted and at which own discard, who won from whom with which delta, draw or not) — no running flags across events;
  * the final scores and ranks do not come from the logs at all but from the arena's own bookkeeping (`Arena.result`,
    game.rs:180-218) — `Stat` re-derives them by adding the logged deltas to the start_kyoku scores (stat.rs:404-437);
  * table-level identities: points sum to zero, ranks form a permutation, wins == hora events, deal-ins == ron events,
    every draw counts for all four seats.

VERDICT r02 weak #2: `Stat` had been checked on one example game only.  CPU only (the oracle is the generator)."""
import numpy as np

import parity_util
from libriichi.stat import Stat


def _play(oracle, n_games, seed0):
    arena = oracle.Arena(parity_util.default_seeds(n_games, seed0), deal_algo=1, enable_quick_eval=True, version=3, keep_log=True)
    d = parity_util.DISCARD_ROW[3]
    for cycle in range(4000):
        rows = arena.poll()
        n = len(rows)
        if n == 0 and arena.n_live == 0:
            break
        obs, masks = arena.encode(0, n, want_obs=True)
        arena.commit(parity_util.greedy_actions(masks, rows, cycle, obs[:, d:d + 3], 0x51A7) if n else np.zeros(0, np.int32))
    assert arena.n_live == 0
    return arena


def _kyokus(events):
    out, cur = [], None
    for ev in events:
        if ev["type"] == "start_kyoku":
            cur = []
            out.append(cur)
        if cur is not None:
            cur.append(ev)
    return out


def _second_reading(events, pid, final_scores):
    """Counters of seat `pid`, kyoku by kyoku."""
    c = dict(round=0, oya=0, fuuro=0, fuuro_num=0, riichi=0, riichi_as_oya=0, riichi_jun=0, chasing_riichi=0, riichi_got_chased=0,
             agari=0, agari_as_oya=0, agari_jun=0, agari_point_oya=0, agari_point_ko=0, riichi_agari=0, fuuro_agari=0, dama_agari=0,
             houjuu=0, houjuu_to_oya=0, houjuu_point_to_oya=0, houjuu_point_to_ko=0, riichi_houjuu=0, fuuro_houjuu=0,
             ryukyoku=0, ryukyoku_point=0, riichi_ryukyoku=0, yakuman=0, nagashi_mangan=0)
    for ky in _kyokus(events):
        oya = ky[0]["oya"]
        c["round"] += 1
        c["oya"] += oya == pid
        types = [e["type"] for e in ky]
        calls = sum(1 for e in ky if e["type"] in ("chi", "pon", "daiminkan") and e["actor"] == pid)
        c["fuuro"] += calls > 0
        c["fuuro_num"] += calls
        reach_idx = {e["actor"]: i for i, e in enumerate(ky) if e["type"] == "reach"}  # one declaration per seat and kyoku
        accepted = any(e["type"] == "reach_accepted" and e["actor"] == pid for e in ky)
        declared = pid in reach_idx
        if declared:
            i = reach_idx[pid]
            c["riichi"] += 1
            c["riichi_as_oya"] += oya == pid
            c["riichi_jun"] += sum(1 for e in ky[:i] if e["type"] == "dahai" and e["actor"] == pid)
            c["chasing_riichi"] += any(j < i for a, j in reach_idx.items() if a != pid)
            c["riichi_got_chased"] += sum(1 for a, j in reach_idx.items() if a != pid and j > i)
        own_discards = sum(1 for e in ky if e["type"] == "dahai" and e["actor"] == pid)
        for e in ky:
            if e["type"] == "hora":
                delta = e["deltas"][pid]
                if e["actor"] == pid:
                    pt = delta - (1000 if accepted else 0)
                    c["agari"] += 1
                    c["agari_jun"] += own_discards
                    c["agari_as_oya"] += oya == pid
                    c["agari_point_oya" if oya == pid else "agari_point_ko"] += pt
                    c["riichi_agari" if accepted else "fuuro_agari" if calls else "dama_agari"] += 1
                    c["yakuman"] += pt >= (48000 if oya == pid else 32000)
                elif e["target"] == pid:
                    c["houjuu"] += 1
                    c["houjuu_to_oya"] += e["actor"] == oya
                    c["houjuu_point_to_oya" if e["actor"] == oya else "houjuu_point_to_ko"] += delta
                    if declared:
                        c["riichi_houjuu"] += 1
                    elif calls:
                        c["fuuro_houjuu"] += 1
            elif e["type"] == "ryukyoku":
                c["ryukyoku"] += 1
                c["ryukyoku_point"] += e["deltas"][pid]
                c["riichi_ryukyoku"] += accepted
                c["nagashi_mangan"] += e["deltas"][pid] >= 8000
        assert types.count("hora") + types.count("ryukyoku") >= 1 and types[-1] == "end_kyoku"
    order = sorted(range(4), key=lambda s: (-int(final_scores[s]), s))
    c["rank"] = order.index(pid) + 1
    c["point"] = int(final_scores[pid]) - 25000
    c["tobi"] = int(final_scores[pid] < 0)
    return c


def test_stat_matches_a_second_reading_of_generated_games(oracle):
    n_games = 16
    arena = _play(oracle, n_games, 97531)
    total = [Stat() for _ in range(4)]
    seen = dict(riichi=0, fuuro=0, ron=0, tsumo=0, ryukyoku=0, chasing=0, tobi=0, oya_agari=0, riichi_ryukyoku=0, double_ron=0)
    for g in range(n_games):
        events = arena.log(g)
        final, done = arena.result(g)
        assert done and int(final.sum()) == 100000
        stats = [Stat.from_game(events, p) for p in range(4)]
        for p in range(4):
            s, c = stats[p], _second_reading(events, p, final)
            for k, v in c.items():
                if k == "rank":
                    assert [s.rank_1, s.rank_2, s.rank_3, s.rank_4] == [int(v == r) for r in (1, 2, 3, 4)], (g, p, v)
                else:
                    assert getattr(s, k) == int(v), (g, p, k, getattr(s, k), v)
            # splits add up (stat.rs:315-345): every win is a riichi, an open or a dama win; points likewise
            assert s.agari == s.riichi_agari + s.fuuro_agari + s.dama_agari
            assert s.agari_point_oya + s.agari_point_ko == s.riichi_agari_point + s.fuuro_agari_point + s.dama_agari_point
            assert s.agari_jun == s.riichi_agari_jun + s.fuuro_agari_jun + s.dama_agari_jun
            assert s.game == 1 and s.rank_1 + s.rank_2 + s.rank_3 + s.rank_4 == 1
            total[p] += s
        # the table as a whole
        horas = [e for e in events if e["type"] == "hora"]
        assert sum(s.point for s in stats) == 0
        assert sorted(s.avg_rank for s in stats) == [1.0, 2.0, 3.0, 4.0]
        assert sum(s.agari for s in stats) == len(horas)
        assert sum(s.houjuu for s in stats) == sum(1 for e in horas if e["actor"] != e["target"])
        n_ryu = sum(1 for e in events if e["type"] == "ryukyoku")
        assert all(s.ryukyoku == n_ryu for s in stats)
        assert sum(s.riichi for s in stats) == sum(1 for e in events if e["type"] == "reach")
        seen["riichi"] += sum(s.riichi for s in stats)
        seen["fuuro"] += sum(s.fuuro for s in stats)
        seen["ron"] += sum(s.houjuu for s in stats)
        seen["tsumo"] += sum(1 for e in horas if e["actor"] == e["target"])
        seen["ryukyoku"] += n_ryu
        seen["chasing"] += sum(s.chasing_riichi for s in stats)
        seen["tobi"] += sum(s.tobi for s in stats)
        seen["oya_agari"] += sum(s.agari_as_oya for s in stats)
        seen["riichi_ryukyoku"] += sum(s.riichi_ryukyoku for s in stats)
        seen["double_ron"] += sum(1 for ky in _kyokus(events) if sum(1 for e in ky if e["type"] == "hora") > 1)
    # the sample exercises the branches it is meant to pin
    assert seen["riichi"] > 40 and seen["fuuro"] > 20 and seen["ron"] > 30 and seen["tsumo"] > 10 and seen["ryukyoku"] > 10, seen
    assert seen["chasing"] > 3 and seen["oya_agari"] > 10 and seen["riichi_ryukyoku"] > 3, seen
    # sums over games (derive_more Add) and the derived rates are plain ratios of the counters
    for p in range(4):
        t = total[p]
        assert t.game == n_games and t.rank_1 + t.rank_2 + t.rank_3 + t.rank_4 == n_games
        assert abs(t.avg_rank - (t.rank_1 + 2 * t.rank_2 + 3 * t.rank_3 + 4 * t.rank_4) / n_games) < 1e-12
        assert abs(t.agari_rate - t.agari / t.round) < 1e-12 and abs(t.riichi_rate - t.riichi / t.round) < 1e-12
        assert abs(t.avg_pt([90, 45, 0, -135]) - (90 * t.rank_1 + 45 * t.rank_2 - 135 * t.rank_4) / n_games) < 1e-12
    assert sum(t.point for t in total) == 0
