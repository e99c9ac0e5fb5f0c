"""Host side of SURVEY §8(f).1: `libriichi.stat.Stat` and the mjai log writer, on the reference's example game
(tests/golden/example_game.jsonl = log-viewer/index.example.html:10-264, metadata stripped)."""
import gzip
import json
import math
import os

import numpy as np

from libriichi.stat import Stat
from mortal_amd import mjai_log

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "example_game.jsonl")


def test_stat_from_log_example_game():
    log = open(GOLDEN).read()
    # seat 1 ("mortal"): pon in E1, chi + ron 5200 in E2, chasing riichi + ron 20000 (incl. its own stick) as dealer in E3
    s = Stat.from_log(log, 1)
    assert (s.game, s.round, s.oya) == (1, 3, 1)
    assert (s.rank_1, s.rank_2, s.rank_3, s.rank_4, s.tobi) == (1, 0, 0, 0, 0)
    assert s.point == 25000 + 5200 + 20000 - 1000 - 25000
    assert (s.agari, s.agari_as_oya, s.agari_point_ko, s.agari_point_oya) == (2, 1, 5200, 19000)
    assert (s.fuuro, s.fuuro_num, s.fuuro_agari, s.fuuro_agari_point, s.fuuro_point) == (2, 2, 1, 5200, 5200)
    assert (s.riichi, s.riichi_as_oya, s.chasing_riichi, s.riichi_agari, s.riichi_agari_point) == (1, 1, 1, 1, 19000)
    assert s.avg_rank == 1.0 and s.total_pt([90, 45, 0, -135]) == 90 and s.avg_pt([90, 45, 0, -135]) == 90.0
    assert math.isnan(s.avg_point_per_dama_agari) and math.isnan(s.houjuu_to_oya_rate)  # 0/0 like Rust's f64
    # seat 2 dealt into all three wins and went below zero
    t = Stat.from_log(log, 2)
    assert (t.houjuu, t.houjuu_to_oya, t.houjuu_point_to_oya, t.houjuu_point_to_ko) == (3, 2, -25700, -4200)
    assert (t.tobi, t.rank_4, t.point) == (1, 1, -30900)
    assert (t.riichi, t.riichi_houjuu, t.fuuro_houjuu, t.riichi_got_chased) == (1, 1, 1, 1)
    # whole table: every round is won by someone, points sum to zero, ranks form a permutation
    allp = [Stat.from_log(log, p) for p in range(4)]
    assert sum(x.point for x in allp) == 0
    assert sum(x.agari for x in allp) == 3 and sum(x.houjuu for x in allp) == 3
    assert sorted(x.avg_rank for x in allp) == [1.0, 2.0, 3.0, 4.0]
    tot = sum(allp, Stat())
    assert tot.game == 4 and tot.round == 12
    text = str(s)
    assert "Avg winning Δscore as dealer     19000.000000" in text and "Avg dama winning Δscore          NaN" in text


def test_stat_from_dir_and_log_writer(tmp_path):
    lines = [json.loads(l) for l in open(GOLDEN)]
    names, seed, events = lines[0]["names"], lines[0]["seed"], lines[1:-1]
    # the writer reproduces the reference's serialisation byte for byte (field order, separators, trailing newline)
    assert mjai_log.dump_json_log(names, seed, events) == open(GOLDEN).read()
    p = mjai_log.write_game_log_as(str(tmp_path / f"{seed[0]}_{seed[1]}_a.json.gz"), names, seed, events)
    assert gzip.open(p, "rt").read() == open(GOLDEN).read()
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "plain.json").write_text(open(GOLDEN).read())
    s = Stat.from_dir(str(tmp_path), "mortal", disable_progress_bar=True)
    assert s.game == 2 and s.rank_1 == 2 and s.point == 2 * 24200  # the gz and the plain copy, seat 1 each
    b = Stat.from_dir(str(tmp_path), "baseline", True)
    assert b.game == 6 and b.point == -2 * 24200
    assert Stat.from_dir(str(tmp_path), "nobody", True).game == 0


def test_decode_events_word_format():
    """Header/payload layout of mortal_amd/csrc/mj_state.h LG_* (the GPU test compares real device logs with the oracle)."""
    def word(t, actor=0, target=0, pai=0, c=(0, 0, 0, 0), tsumogiri=0):
        return t | actor << 4 | target << 6 | pai << 8 | c[0] << 14 | c[1] << 20 | c[2] << 26 | c[3] << 32 | tsumogiri << 38

    def i32x2(a, b):
        return (a & 0xFFFFFFFF) | ((b & 0xFFFFFFFF) << 32)

    tiles = [i % 37 for i in range(52)]
    hai = [sum((tiles[w * 8 + k] if w * 8 + k < 52 else 0) << (8 * k) for k in range(8)) for w in range(7)]
    words = [word(1, pai=29, c=(5, 0, 0, 0)) | 2 << 44 | 3 << 52, i32x2(25000, 24000), i32x2(-100, 51100), *hai,
             word(2, actor=1, pai=34), word(3, actor=1, pai=34, tsumogiri=1), word(4, 2, 1, 4, (34, 5, 0, 0)),
             word(8, actor=3, c=(35, 13, 13, 13)), word(9, pai=31), word(7, actor=0, pai=36, c=(22, 22, 22, 0)),
             word(12, 2, 0) | 2 << 39, i32x2(8000, -8000), i32x2(0, 0), 7 | 8 << 6,
             word(13), i32x2(1500, -1500), i32x2(1500, -1500), word(14)]
    ev = mjai_log.decode_events(np.array(words, dtype=np.uint64))
    assert ev[0] == {"type": "start_kyoku", "bakaze": "S", "dora_marker": "W", "kyoku": 2, "honba": 2, "kyotaku": 3,
                     "oya": 1, "scores": [25000, 24000, -100, 51100],
                     "tehais": [[mjai_log.TILE_NAMES[t] for t in tiles[s * 13:(s + 1) * 13]] for s in range(4)]}
    assert ev[1:] == [
        {"type": "tsumo", "actor": 1, "pai": "5mr"}, {"type": "dahai", "actor": 1, "pai": "5mr", "tsumogiri": True},
        {"type": "chi", "actor": 2, "target": 1, "pai": "5m", "consumed": ["5mr", "6m"]},
        {"type": "ankan", "actor": 3, "consumed": ["5pr", "5p", "5p", "5p"]}, {"type": "dora", "dora_marker": "P"},
        {"type": "kakan", "actor": 0, "pai": "5sr", "consumed": ["5s", "5s", "5s"]},
        {"type": "hora", "actor": 2, "target": 0, "deltas": [8000, -8000, 0, 0], "ura_markers": ["8m", "9m"]},
        {"type": "ryukyoku", "deltas": [1500, -1500, 1500, -1500]}, {"type": "end_kyoku"}]
    assert list(ev[0].keys()) == ["type", "bakaze", "dora_marker", "kyoku", "honba", "kyotaku", "oya", "scores", "tehais"]


def test_encode_decode_roundtrip_and_augment():
    """encode_events / decode_events are inverse on the example game (incl. the replay-only wall payload), and the suit
    augmentation (Tile::augment, tile.rs:154-167) is an involution that swaps manzu and pinzu only."""
    lines = [json.loads(l) for l in open(GOLDEN)]
    events = lines[1:-1]
    words = mjai_log.encode_events(lines)  # start_game / end_game are skipped
    assert mjai_log.decode_events(words) == events
    aug = mjai_log.decode_events(mjai_log.encode_events(events, augmented=True))
    assert aug != events and mjai_log.decode_events(mjai_log.encode_events(aug, augmented=True)) == events
    swap = {"m": "p", "p": "m"}
    t0, t1 = events[0]["tehais"][0][0], aug[0]["tehais"][0][0]
    assert t1 == (t0[0] + swap.get(t0[1], t0[1]) + t0[2:] if len(t0) > 1 and t0[1] in "mps" else t0)
    for t in range(37):
        assert mjai_log.augment_tile_id(mjai_log.augment_tile_id(t)) == t
    assert [mjai_log.augment_tile_id(t) for t in (0, 9, 18, 27, 34, 35, 36)] == [9, 0, 18, 27, 35, 34, 36]


def test_walls_from_events_are_consistent():
    """GameplayLoader's wall reconstruction for oracle=True without a seed (dataset/invisible.rs:73-149): every kyoku's wall
    is a permutation of the 136 tiles, starts with the logged haipai, replays the logged draws / dora / ura in order."""
    from mortal_amd.dataset import GameplayLoader

    events = [json.loads(l) for l in open(GOLDEN)]
    walls = GameplayLoader._walls_from_events(events, False, np.random.default_rng(0))
    assert len(walls) == 3
    full = sorted([t for t in range(34) for _ in range(4)])
    deaka = {34: 4, 35: 13, 36: 22}
    tid = mjai_log.TILE_ID
    kyokus = []
    for e in events:
        if e["type"] == "start_kyoku":
            kyokus.append([e])
        elif kyokus and e["type"] != "end_game":
            kyokus[-1].append(e)
    for wall, kev in zip(walls, kyokus):
        assert sorted(deaka.get(t, t) for t in wall) == full and sorted(t for t in wall if t >= 34) == [34, 35, 36]
        assert wall[:52] == [tid[x] for h in kev[0]["tehais"] for x in h]
        assert wall[60] == tid[kev[0]["dora_marker"]]
        draws = [tid[e["pai"]] for e in kev if e["type"] == "tsumo"]  # no kans in this game: all from the yama
        assert [wall[66 + 69 - k] for k in range(len(draws))] == draws
        ura = next(e["ura_markers"] for e in kev if e["type"] == "hora")
        assert [wall[61 + k] for k in range(len(ura))] == [tid[x] for x in ura]
    # the words carry the wall and still decode to the same events
    w = mjai_log.encode_events(events, walls=walls)
    assert mjai_log.decode_events(w) == events[1:-1]


EVENT_LINES = os.path.join(os.path.dirname(__file__), "golden", "event_lines.jsonl")


def test_event_lines_of_the_reference_round_trip(oracle):
    """mjai/event.rs:260-294 (`json_consistency`): every event kind, with and without the optional hora / ryukyoku
    fields.  The oracle's packed event and the device's LG_* word format both reproduce the reference's serialisation
    (serde declaration order, compact separators) byte for byte."""
    lines = [l.rstrip("\n") for l in open(EVENT_LINES)]
    assert len(lines) == 19
    kinds = set()
    for line in lines:
        ev = json.loads(line)
        kinds.add(ev["type"])
        if ev["type"] not in ("none", "start_game", "end_game"):  # written by dump_json_log / never logged
            assert mjai_log._dumps(oracle.unpack_event(oracle.pack_event(ev))) == line
        if ev["type"] in ("none", "start_game", "end_game"):
            assert len(mjai_log.encode_events([ev])) == 0
            continue
        words = mjai_log.encode_events([ev])
        got = mjai_log.decode_events(words)
        assert len(got) == 1
        if ev["type"] in ("hora", "ryukyoku") and "deltas" not in ev:
            # the word format always carries the deltas (the arena always logs them): absent == zeros
            want = dict(ev, deltas=[0, 0, 0, 0])
            if ev["type"] == "hora":
                want["ura_markers"] = []
            assert got[0] == want
        else:
            assert mjai_log._dumps(got[0]) == line
    assert kinds == {"none", "start_game", "start_kyoku", "tsumo", "dahai", "chi", "pon", "daiminkan", "kakan", "ankan", "dora",
                     "reach", "reach_accepted", "hora", "ryukyoku", "end_kyoku", "end_game"}
    names, seed = json.loads(lines[1])["names"], json.loads(lines[1])["seed"]
    assert mjai_log.dump_json_log(names, seed, []).splitlines() == [lines[1], lines[-1]]


def test_event_bound_check():
    """mjai/event.rs:296-337 (`bound_check`): actor / target above 3 and kyoku outside 1..=4 are errors, not wrapped."""
    import pytest

    sk = json.loads(open(EVENT_LINES).read().splitlines()[2])
    assert len(mjai_log.encode_events([sk])) == 10
    for bad in ({"type": "reach", "actor": 4}, {"type": "hora", "actor": 0, "target": 5}, dict(sk, kyoku=0), dict(sk, kyoku=5),
                dict(sk, oya=4)):
        with pytest.raises(ValueError):
            mjai_log.encode_events([bad])
