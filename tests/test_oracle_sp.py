"""Oracle SP calculator vs the reference's KATs (algo/sp/calc.rs:773-1007; f32 equality within f32::EPSILON,
calc.rs:769-771).  These run with calc_tegawari=calc_shanten_down=true, a superset of the production flags
(state/agent_helper.rs:581-584)."""
import numpy as np

EPS = float(np.finfo(np.float32).eps)
E, S, W, N, P, F, C = range(27, 34)


def feq(a, b):
    return abs(np.float32(a) - np.float32(b)) <= EPS


def _run(o, hand_s, dora, **kw):
    tehai = o.hand(hand_s)
    seen = tehai.copy()
    for d in dora:
        seen[d] += 1
    extra_seen = kw.pop("extra_seen", {})
    for t, c in extra_seen.items():
        seen[t] += c
    cur = o.calc_shanten(tehai, 4)
    return o.sp_calc(tehai, seen, dora_indicators=dora, cur_shanten=cur, calc_tegawari=True, calc_shanten_down=True,
                     **kw), seen


def test_nanikiru_1(oracle):  # calc.rs:794-816
    c, _ = _run(oracle, "45678m 34789p 3344z", [P], jikaze=N, tsumos_left=8)
    assert c[0]["tile"] == N and c[1]["tile"] == W
    assert tuple(c[0]["exp_values"]) > tuple(c[1]["exp_values"])


def test_nanikiru_2(oracle):  # calc.rs:820-846
    c, _ = _run(oracle, "3667m 23489p 34688s", [P], jikaze=N, tsumos_left=15)
    assert c[0]["tile"] == oracle.TILE_ID["9p"] and c[0]["shanten_down"]
    c, _ = _run(oracle, "3667m 23489p 34688s", [P], jikaze=N, tsumos_left=15, maximize_win_prob=True)
    assert c[0]["tile"] == oracle.TILE_ID["3m"] and not c[0]["shanten_down"]


def test_nanikiru_3(oracle):  # calc.rs:850-900
    c, _ = _run(oracle, "45677m 456778p 248s", [oracle.TILE_ID["6m"]], jikaze=E, tsumos_left=15,
                calc_double_riichi=True, calc_haitei=True)
    c = c[0]
    assert c["tile"] == oracle.TILE_ID["2s"]
    assert len(c["required_tiles"]) == 17 and c["num_required_tiles"] == 57 and c["shanten_down"]
    assert feq(c["tenpai_probs"][0], 0.90023905), c["tenpai_probs"][0]
    assert feq(c["win_probs"][0], 0.34794784), c["win_probs"][0]
    assert feq(c["exp_values"][0], 5894.7617), c["exp_values"][0]


def test_nanikiru_4(oracle):  # calc.rs:904-949
    c, seen = _run(oracle, "9999m 6677p 88s 335z 1m", [oracle.TILE_ID["1m"]], jikaze=W, tsumos_left=5)
    assert len(c) == 7
    c = c[1]
    assert c["tile"] == oracle.TILE_ID["1m"] and c["shanten_down"]
    assert len(c["required_tiles"]) == 33
    assert c["num_required_tiles"] == 34 * 4 - int(seen.sum())


def test_tsumo_only(oracle):  # calc.rs:953-1006
    c, _ = _run(oracle, "45677m 456778p 48s", [oracle.TILE_ID["6m"]], jikaze=W, tsumos_left=5, can_discard=False,
                calc_double_riichi=True, calc_haitei=True, maximize_win_prob=True,
                extra_seen={oracle.TILE_ID["5s"]: 4}, akas_seen=(0, 0, 1))
    assert len(c) == 1
    c = c[0]
    assert c["tile"] == 37
    assert len(c["required_tiles"]) == 16 and c["num_required_tiles"] == 54
    assert feq(c["tenpai_probs"][0], 0.45017204), c["tenpai_probs"][0]
    assert feq(c["win_probs"][0], 0.03441279), c["win_probs"][0]
    assert feq(c["exp_values"][0], 432.26678), c["exp_values"][0]
