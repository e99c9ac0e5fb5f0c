"""A SECOND reading of the obs layout (SURVEY.md Appendix C row budget), independent of oracle/obs.cc's running index.

The obs tensor values have no reference vectors, and oracle/obs.cc is one line-by-line restatement of
state/obs_repr.rs:126-630.  This test re-derives every block offset of every version from the Appendix C table (block widths
per version, written down separately from the encoder) and checks, on a few hundred real decision states of a tenpai-seeking
game, that the planes whose meaning is directly observable in the PlayerState — hand, akas, scores, rank, round wind / seat
wind, tiles left, waits, furiten, shanten, riichi flags, the discard block, the action rows — sit at those offsets with those
values, and that the 46-wide mask agrees with the planes.  The device encoder equals the oracle bit for bit (-m gpu and the
emulator tests), so this pins both against a second interpretation of the row map."""
import numpy as np
import pytest

import parity_util

# Appendix C: (block name, widths for v1, v2, v3, v4) in encoding order
BLOCKS = [
    ("hand", 4, 4, 4, 4), ("akas", 3, 3, 3, 3), ("scores", 4, 40, 40, 8), ("rank", 4, 4, 4, 4), ("kyoku", 4, 4, 4, 4),
    ("honba_kyotaku", 20, 4, 4, 2), ("bakaze_jikaze", 2, 2, 2, 2), ("kyoku_in_game", 0, 1, 1, 1), ("dora_indicators", 7, 7, 7, 7),
    ("own_kawa", 96, 96, 96, 96), ("own_kawa_decay", 0, 0, 1, 1), ("opp_kawa", 576, 576, 576, 576), ("opp_extra", 0, 18, 9, 9),
    ("tiles_left", 1, 1, 1, 1), ("doras_owned", 48, 12, 12, 4), ("doras_unseen", 23, 4, 4, 1), ("kawa_overview", 28, 28, 28, 28),
    ("fuuro_overview", 80, 80, 80, 80), ("ankan_overview", 4, 4, 4, 4), ("seen_tedashi_riichi_tiles", 0, 19, 19, 19),
    ("riichi_declared_accepted", 6, 6, 6, 6), ("waits", 1, 1, 1, 1), ("furiten", 1, 1, 1, 1), ("shanten", 6, 7, 7, 7),
    ("self_riichi_accepted", 1, 1, 1, 1), ("kan_select", 1, 1, 1, 1), ("target_tile", 3, 3, 3, 3), ("discard", 5, 5, 5, 5),
    ("actions", 10, 10, 10, 10), ("sp", 0, 0, 0, 123),
]
TOTAL = {1: 938, 2: 942, 3: 934, 4: 1012}  # consts.rs:22-25


def offsets(version):
    off, out = 0, {}
    for name, *w in BLOCKS:
        out[name] = (off, w[version - 1])
        off += w[version - 1]
    assert off == TOTAL[version]
    return out


def test_row_budget_matches_the_constants_used_elsewhere():
    for v in (1, 2, 3, 4):
        o = offsets(v)
        assert o["discard"][0] == parity_util.DISCARD_ROW[v]  # the tests' / device policy's discard-block anchor
    assert offsets(4)["sp"][0] == 889  # mj_sp.hip O_SP


@pytest.mark.parametrize("version", [1, 2, 3, 4])
def test_observable_planes_sit_where_appendix_c_says(oracle, version):
    o = offsets(version)
    seeds = parity_util.default_seeds(6, 4321)
    arena = oracle.Arena(seeds, deal_algo=1, enable_quick_eval=False, version=version, keep_log=False)
    checked = riichi_rows = furiten_rows = call_rows = 0
    deaka = np.arange(37)
    deaka[34:] = [4, 13, 22]
    ones, zeros = np.ones(34, dtype=np.float32), np.zeros(34, dtype=np.float32)
    for cycle in range(500):
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
            x, m = obs[r], masks[r].astype(bool)
            assert x.shape[0] == TOTAL[version] and x.min() >= 0.0 and x.max() <= 1.0
            h0 = o["hand"][0]
            for k in range(4):  # thermometer over the count of each tile (obs_repr.rs:130-140)
                assert (x[h0 + k] == (sn["tehai"] > k)).all()
            a0 = o["akas"][0]
            for i in range(3):
                assert (x[a0 + i] == (ones if sn["akas_in_hand"][i] else zeros)).all()
            s0, sw = o["scores"]
            per = sw // 4
            for i in range(4):  # scores relative to the seat, first row of each = score / 100k (obs_repr.rs:149-165)
                want = np.float32(min(max(sn["scores"][i], 0), 100000)) / np.float32(100000.0)
                assert (x[s0 + i * per] == want).all()
                if version == 4:
                    want30 = np.float32(min(max(sn["scores"][i], 0), 30000)) / np.float32(30000.0)
                    assert (x[s0 + i * per + 1] == want30).all()
            r0 = o["rank"][0]
            assert [bool(x[r0 + k].all()) for k in range(4)] == [k == sn["rank"] for k in range(4)]
            assert (x[o["tiles_left"][0]] == np.float32(sn["tiles_left"]) / np.float32(69.0)).all()
            assert (x[o["waits"][0]] == sn["waits"]).all()
            assert (x[o["furiten"][0]] == (ones if sn["at_furiten"] else zeros)).all()
            furiten_rows += sn["at_furiten"]
            sh0, shw = o["shanten"]
            if version == 1:  # thermometer over 6 rows
                assert [bool(x[sh0 + k].all()) for k in range(6)] == [k < sn["shanten"] for k in range(6)]
            else:  # one-hot over 0..6
                assert [bool(x[sh0 + k].all()) for k in range(7)] == [k == sn["shanten"] for k in range(7)]
            assert not x[o["kan_select"][0]].any()
            cans = sn["cans"]
            d0 = o["discard"][0]
            if cans["can_discard"]:
                legal = np.zeros(34, dtype=bool)
                legal[deaka[np.flatnonzero(m[:37])]] = True
                assert (x[d0] == legal).all()                           # discard candidates == the mask's tiles
                assert (x[d0 + 1] == sn["keep_shanten_discards"]).all()
                assert (x[d0 + 2] == sn["next_shanten_discards"]).all()
            else:
                assert not x[d0:d0 + 5].any() and not m[:37].any()
            a0 = o["actions"][0]  # riichi 1, chi 3, pon 1, daiminkan 1, ankan 1, kakan 1, agari 1, ryukyoku 1 (obs_repr.rs:478-562)
            flags = [cans["can_riichi"], cans["can_chi_low"], cans["can_chi_mid"], cans["can_chi_high"], cans["can_pon"],
                     cans["can_daiminkan"], cans["can_ankan"], cans["can_kakan"], cans["can_tsumo_agari"] or cans["can_ron_agari"],
                     cans["can_ryukyoku"]]
            got = [bool(x[a0 + k].any()) for k in range(10)]
            assert got == [bool(f) for f in flags], (got, flags)
            assert m[37] == bool(cans["can_riichi"]) and m[38:41].tolist() == [bool(f) for f in flags[1:4]]
            assert m[41] == bool(cans["can_pon"]) and m[42] == bool(cans["can_daiminkan"] or cans["can_ankan"] or cans["can_kakan"])
            assert m[43] == bool(flags[8]) and m[44] == bool(cans["can_ryukyoku"])
            can_pass = any(cans[k] for k in ("can_chi_low", "can_chi_mid", "can_chi_high", "can_pon", "can_daiminkan", "can_ron_agari"))
            assert m[45] == can_pass
            t0 = o["target_tile"][0]
            assert bool(x[t0].any()) == can_pass and (not can_pass or x[t0].sum() == 1.0)
            riichi_rows += cans["can_riichi"]
            call_rows += can_pass
            checked += 1
        d = parity_util.DISCARD_ROW[version]
        act = parity_util.greedy_actions(masks, rows, cycle, obs[:, d:d + 3], 0x9E3779B97F4A7C15) if n else np.zeros(0, np.int32)
        arena.commit(act)
    assert checked > 1500 and riichi_rows > 5 and call_rows > 100
