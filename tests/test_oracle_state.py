"""Oracle PlayerState pinned on the reference's own scenario tests (libriichi/src/state/test.rs:71-1418).

tests/golden/state_scenarios.json holds the vectors (events, hands, expected values) extracted by
tests/golden/extract_state_scenarios.py; every Rust assertion text is evaluated here against the oracle through a
small pattern table.  An assertion the table does not understand fails the test (no silent skips).
"""
import json
import os
import re

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
SCEN = json.load(open(os.path.join(HERE, "golden", "state_scenarios.json")))

TID = O.TILE_ID


def _tiles(text):
    return [TID[x.strip()] for x in text.split(",") if x.strip()]


class Runner:
    def __init__(self):
        self.vars = {}  # rust variable -> PlayerState
        self.cans = {}  # rust binding -> cans dict
        self.expected = None
        self.discard_candidates = None
        self.rank = None
        self.should_hora = None
        self.saved_scores = None
        self.n_asserts = 0

    def ps(self, name):
        return self.vars[name]

    def check(self, cond, text):
        self.n_asserts += 1
        assert cond, text

    def step(self, s):
        if "new" in s:
            self.vars[s["var"]] = O.PlayerState(s["new"])
        elif "ev" in s:
            cans = self.ps(s["on"]).update(s["ev"])
            if s.get("bind"):
                self.cans[s["bind"]] = cans
        elif "set_tehai" in s:
            ps = self.ps(s["on"])
            n = s["len_div3"] if s["len_div3"] is not None else ps.snapshot()["tehai_len_div3"]
            ps.set_tehai(O.hand(s["set_tehai"]), n)
        elif "call" in s:
            ps = self.ps(s["on"])
            if s["call"] == "update_waits_and_furiten":
                ps.call(0)
            else:
                ps.call(1, TID[s["arg"]])
        elif "clone" in s:
            self.vars[s["var"]] = self.ps(s["clone"]).clone()
        else:
            self.eval(s["assert"])

    # ---- the assertion pattern table
    def eval(self, a):
        m = re.fullmatch(r"assert!\((\w+)\.shanten == (-?\d+)\);", a) or re.fullmatch(r"assert_eq!\((\w+)\.shanten, (-?\d+)\);", a)
        if m:
            return self.check(self.ps(m[1]).snapshot()["shanten"] == int(m[2]), a)
        m = re.fullmatch(r"assert_eq!\((\w+)\.real_time_shanten\(\), (-?\d+)\);", a)
        if m:
            return self.check(self.ps(m[1]).snapshot()["real_time_shanten"] == int(m[2]), a)
        m = re.fullmatch(r"assert!\((\w+)\.waits\.iter\(\)\.all\(\|&b\| !b\)\);", a)
        if m:
            return self.check(not self.ps(m[1]).snapshot()["waits"].any(), a)
        m = re.fullmatch(r"assert!\(((?:\w+\.waits\[[^\]]+\](?: && )?)+)\);", a)
        if m:
            w = None
            for name, idx in re.findall(r"(\w+)\.waits\[([^\]]+)\]", m[1]):
                w = self.ps(name).snapshot()["waits"]
                mm = re.fullmatch(r"tuz!\((\w+)\)", idx)
                i = TID[mm[1]] if mm else int(idx)
                self.check(bool(w[i]), a)
            return
        m = re.fullmatch(r"assert!\((!?)(\w+)\.(at_furiten|has_next_shanten_discard)\);", a)
        if m:
            return self.check(self.ps(m[2]).snapshot()[m[3]] == (m[1] == ""), a)
        m = re.fullmatch(r"assert!\((!?)(\w+)\.last_cans\.(\w+)\);", a)
        if m:
            return self.check(bool(self.ps(m[2]).snapshot()["cans"][m[3]]) == (m[1] == ""), a)
        m = re.fullmatch(r"assert!\((!?)(\w+)\.(can_\w+)\);", a)
        if m and m[2] in self.cans:
            return self.check(bool(self.cans[m[2]][m[3]]) == (m[1] == ""), a)
        m = re.fullmatch(r"assert!\(matches!\( (\w+)\.last_cans, ActionCandidate \{ (.*), \.\. \}, \)\);", a)
        if m:
            c = self.ps(m[1]).snapshot()["cans"]
            for k, v in re.findall(r"(\w+): (true|false)", m[2]):
                self.check(bool(c[k]) == (v == "true"), a)
            return
        m = re.fullmatch(r"assert_eq!\((\w+)\.doras_owned\[0\], (\d+)\);", a)
        if m:
            return self.check(self.ps(m[1]).snapshot()["doras_owned"][0] == int(m[2]), a)
        m = re.fullmatch(r"assert_eq!\((\w+)\.agari_points\((true|false), &\[(.*)\]\)\.unwrap\(\)\.(\w+), (\d+)\);", a)
        if m:
            ura = [x for x in re.findall(r"t!\(([^)]+)\)", m[3])]
            p = self.ps(m[1]).agari_points(m[2] == "true", ura)
            return self.check(p[m[4]] == int(m[5]), a + f" got {p}")
        m = re.fullmatch(r"let expected = t!\[(.*)\];", a)
        if m:
            self.expected = set(_tiles(m[1]))
            return
        if a.startswith("for (idx, &b) in ps.waits.iter().enumerate()"):
            w = self.ps("ps").snapshot()["waits"]
            return self.check({i for i in range(34) if w[i]} == self.expected, a)
        if a.startswith("ps.discard_candidates_with_unconditional_tenpai() .iter()"):
            d = self.ps("ps").uncond_tenpai()
            return self.check({i for i in range(34) if d[i]} == self.expected, a + f" got {np.nonzero(d)[0]}")
        if a == "let discard_candidates = ps.discard_candidates_with_unconditional_tenpai();":
            self.discard_candidates = self.ps("ps").uncond_tenpai()
            return
        if a == "assert_eq!(discard_candidates, [false; 34]);":
            return self.check(not self.discard_candidates.any(), a)
        m = re.fullmatch(r"let orig_scores = mem::replace\(&mut ps\.scores, \[(.*)\]\);", a)
        if m:
            self.saved_scores = self.ps("ps").snapshot()["scores"]
            self.ps("ps").set_scores([int(x) for x in m[1].split(",")])
            return
        if a == "ps.scores = orig_scores;":
            self.ps("ps").set_scores(self.saved_scores)
            return
        m = re.fullmatch(r"ps\.add_dora_indicator\(t!\((\w+)\)\)\.unwrap\(\);", a)
        if m:
            self.ps("ps").call(5, TID[m[1]])
            return
        m = re.fullmatch(r"let rank = ps\.get_rank\(\[(.*)\]\);", a)
        if m:
            self.rank = self.ps("ps").get_rank([int(x) for x in m[1].split(",")])
            return
        m = re.fullmatch(r"assert_eq!\(rank, (\d)\);", a)
        if m:
            return self.check(self.rank == int(m[1]), a + f" got {self.rank}")
        if a == "let should_hora = ps.rule_based_agari();":
            self.should_hora = bool(self.ps("ps").call(2))
            return
        m = re.fullmatch(r"assert!\((!?)should_hora\);", a)
        if m:
            return self.check(self.should_hora == (m[1] == ""), a)
        m = re.fullmatch(r"assert!\((!?)ps\.rule_based_agari\(\)\);", a)
        if m:
            return self.check(bool(self.ps("ps").call(2)) == (m[1] == ""), a)
        raise AssertionError(f"unhandled reference assertion: {a}")


def _invariants(ps):
    """state/test.rs:49-67 `validate`: cheap internal-consistency checks after every event."""
    s = ps.snapshot()
    assert s["tehai"].sum() + 0 in (3 * s["tehai_len_div3"] + 1, 3 * s["tehai_len_div3"] + 2) or s["tehai"].sum() == 0
    assert (s["tiles_seen"] <= 4).all()
    assert (s["tehai"] <= s["tiles_seen"]).all() or s["tehai"].sum() == 0
    assert not (s["next_shanten_discards"] & s["keep_shanten_discards"]).any()
    assert s["has_next_shanten_discard"] == bool(s["next_shanten_discards"].any())


@pytest.mark.parametrize("name", sorted(SCEN))
def test_state_scenario(oracle, name):
    sc = SCEN[name]
    r = Runner()
    started = False
    for s in sc["steps"]:
        r.step(s)
        if "ev" in s:
            started = started or s["ev"]["type"] == "start_kyoku"
            if started:
                _invariants(r.ps(s["on"]))
    assert r.n_asserts > 0, f"{name} (test.rs:{sc['line']}) evaluated no assertions"


def test_scenarios_cover_reference_file():
    # the 10 #[test] functions of state/test.rs at the pinned reference revision
    assert len(SCEN) == 10
    assert sum(sum("assert" in s for s in t["steps"]) for t in SCEN.values()) >= 90


def test_reference_bench_kyoku_replays(oracle):
    """benches/bench.rs:136-241: the reference's encode-obs benchmark input replays on the oracle for every seat, the
    PlayerState invariants hold after each event, and all four obs versions encode."""
    evs = [json.loads(l) for l in open(os.path.join(HERE, "golden", "bench_kyoku.jsonl"))]
    assert evs[0]["type"] == "start_kyoku" and len(evs) > 80
    for pid in range(4):
        ps = O.PlayerState(pid)
        n_act = 0
        for ev in evs:
            cans = ps.update(ev)
            _invariants(ps)
            n_act += any(v for k, v in cans.items() if k != "target_actor")
        assert n_act >= 10
        for v in (1, 2, 3, 4):
            obs, mask = ps.encode_obs(v, False)
            assert obs.shape[1] == 34 and np.isfinite(obs).all() and obs.min() >= 0.0
