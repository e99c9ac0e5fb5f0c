"""`engine_type == 'mjai-log'` (agent/py_agent.rs:24-37 -> agent/mjai_log.rs:12-150).  The reference's own
`ExampleMjaiLogEngine` (mortal/engine.py:96-140, restated below; the reference file is only READ for a textual comparison) plays the challenger of
OneVsThree and one side of TwoVsTwo against a 'mortal'-type engine; its reactions are explicit mjai events applied by the
step kernel (mj_step_ev).  Checked against the oracle arena driven by the same two policies, and the engine's callbacks
(set_player_ids / start_game / end_kyoku / end_game, GameState fields) against the reference's planning.
CPU: the device kernels run on the host emulator (tests/host/emu)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
REF = "/root/reference/mortal"
KEY = 0xD5DFAA4CEF265CD7
if HOST not in sys.path:
    sys.path.insert(0, HOST)


def _example_engine_cls():
    """Always the in-file restatement (the same engine on every host, no external code executed); when the reference is
    present, test_example_engine_restatement_matches_the_reference compares the two texts."""

    class ExampleMjaiLogEngine:  # mortal/engine.py:96-140
        def __init__(self, name):
            self.engine_type = "mjai-log"
            self.name = name
            self.player_ids = None

        def set_player_ids(self, player_ids):
            self.player_ids = player_ids

        def react_batch(self, game_states):
            res = []
            for gs in game_states:
                events = json.loads(gs.events_json)
                assert events[0]["type"] == "start_kyoku"
                if gs.state.last_cans.can_discard:
                    res.append(json.dumps({"type": "dahai", "actor": self.player_ids[gs.game_index],
                                           "pai": gs.state.last_self_tsumo(), "tsumogiri": True}))
                else:
                    res.append('{"type":"none"}')
            return res

        def start_game(self, game_idx):
            pass

        def end_kyoku(self, game_idx):
            pass

        def end_game(self, game_idx, scores):
            pass

    return ExampleMjaiLogEngine


class _Lowest:
    engine_type = "mortal"
    is_oracle = False
    version = 3
    enable_quick_eval = True
    enable_rule_based_agari_guard = False
    name = "lowest"

    def react_batch(self, obs, masks, invisible_obs):
        import torch

        m = torch.as_tensor(np.stack(masks, axis=0))
        return m.to(torch.uint8).argmax(dim=1).tolist(), torch.zeros(m.shape).tolist(), m.tolist(), [True] * m.shape[0]


def _oracle_scores(oracle, seeds, is_logger, deal_algo):
    """The same two policies on the oracle: tsumogiri (discard the drawn tile when a discard is due, else pass) for the
    mjai-log seats, lowest legal action for the others."""
    arena = oracle.Arena(seeds, deal_algo=deal_algo, enable_quick_eval=True, version=3, keep_log=False)
    while arena.n_live > 0:
        rows = arena.poll()
        _, masks = arena.encode(0, len(rows), want_obs=False)
        act = np.zeros(len(rows), dtype=np.int32)
        for r, (g, seat, kan) in enumerate(rows):
            if is_logger(int(g), int(seat)):
                assert not kan
                act[r] = arena.player_state(int(g), int(seat)).snapshot()["last_self_tsumo"] if masks[r, :37].any() else 45
                assert masks[r, act[r]]
            else:
                act[r] = int(masks[r].argmax())
        arena.commit(act)
    return np.array([arena.result(g)[0] for g in range(len(seeds))])


@pytest.fixture(scope="module")
def emu_arena():
    import emu_pool

    from mortal_amd import arena as A

    old = A.BatchRunner.pool_cls
    A.BatchRunner.pool_cls = emu_pool.make_pool_class()
    yield A
    A.BatchRunner.pool_cls = old


def test_mjai_log_challenger_one_vs_three(oracle, emu_arena):
    from libriichi.arena import OneVsThree

    from mortal_amd.pool import default_deal_algo

    calls = {"start": [], "end_kyoku": 0, "end_game": []}
    Base = _example_engine_cls()

    class Eng(Base):
        def start_game(self, idx):
            calls["start"].append(idx)

        def end_kyoku(self, idx):
            calls["end_kyoku"] += 1

        def end_game(self, idx, scores):
            calls["end_game"].append((idx, list(scores)))

    chal = Eng("logger")
    got = OneVsThree(disable_progress_bar=True).py_vs_py(challenger=chal, champion=_Lowest(), seed_start=(10000, KEY), seed_count=1)
    seeds = [(10000, KEY)] * 4
    sc = _oracle_scores(oracle, seeds, lambda g, s: s == g % 4, default_deal_algo())
    want = [0, 0, 0, 0]
    for g in range(4):
        order = sorted(range(4), key=lambda i: -int(sc[g][i]))
        want[order.index(g % 4)] += 1
    assert got == want
    assert chal.player_ids == [0, 1, 2, 3]  # one_vs_three.rs:144
    assert calls["start"] == [0, 1, 2, 3] and calls["end_kyoku"] >= 4 * 4
    assert sorted(calls["end_game"]) == [(g, [int(x) for x in sc[g]]) for g in range(4)]


def test_mjai_log_two_vs_two_and_validation(oracle, emu_arena):
    from libriichi.arena import TwoVsTwo

    from mortal_amd.arena import MortalAmdError  # the class object the arena raises (test_host reloads mortal_amd._lib)
    from mortal_amd.pool import default_deal_algo

    Base = _example_engine_cls()
    eng = Base("logger")
    env = TwoVsTwo(disable_progress_bar=True)
    assert env.py_vs_py(challenger=eng, champion=_Lowest(), seed_start=(20000, KEY), seed_count=1) is None
    assert eng.player_ids == [0, 2, 1, 3]  # two_vs_two.rs:142-172: split A seats 0,2; split B seats 1,3
    sc = _oracle_scores(oracle, [(20000, KEY)] * 2, lambda g, s: (s % 2 == 0) == (g % 2 == 0), default_deal_algo())
    assert (env.last_scores == sc).all()

    class Bad(Base):  # an event the state does not allow: the run aborts with the offending event and the state
        def react_batch(self, game_states):
            return [json.dumps({"type": "dahai", "actor": self.player_ids[gs.game_index], "pai": "5mr", "tsumogiri": False})
                    for gs in game_states]

    with pytest.raises(MortalAmdError, match="invalid action"):
        TwoVsTwo(disable_progress_bar=True).py_vs_py(challenger=Bad("bad"), champion=_Lowest(), seed_start=(20000, KEY), seed_count=1)


def test_unknown_engine_type_is_rejected():
    from mortal_amd.arena import _check_engine

    class E:
        engine_type = "akochan"

    with pytest.raises(ValueError, match="unknown engine type"):
        _check_engine(E())


def test_example_engine_restatement_matches_the_reference():
    """The restated ExampleMjaiLogEngine follows mortal/engine.py:96-140 statement for statement (text comparison only: the
    reference module is never imported or executed)."""
    path = os.path.join(REF, "engine.py")
    if not os.path.exists(path):
        pytest.skip("reference not present on this host")
    src = open(path).read()
    body = src[src.index("class ExampleMjaiLogEngine"):]
    for needle in ("self.engine_type = 'mjai-log'", "def set_player_ids(self, player_ids", "def react_batch(self, game_states)",
                   "assert events[0]['type'] == 'start_kyoku'", "cans = state.last_cans", "if cans.can_discard:",
                   "tile = state.last_self_tsumo()", "'type': 'dahai'", "'actor': player_id", "'tsumogiri': True",
                   "res.append('{\"type\":\"none\"}')", "def start_game(self, game_idx", "def end_kyoku(self, game_idx",
                   "def end_game(self, game_idx"):
        assert needle in body, needle
