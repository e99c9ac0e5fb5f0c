"""libriichi.dataset — Grp on the host, GameplayLoader by log replay on the device."""
import json
import os

import numpy as np
import pytest

import parity_util

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "example_game.jsonl")


def test_grp_example_game():
    from libriichi.dataset import Grp

    g = Grp.load_log(open(GOLDEN).read())
    f = g.take_feature()
    assert f.dtype == np.float64 and f.shape == (3, 7)
    # E1-0, E1-1 (dealer won), E2-0; scores before each kyoku; the riichi stick of E1 went back to its owner with the win
    assert f[:, 0].tolist() == [0.0, 0.0, 1.0] and f[:, 1].tolist() == [0.0, 1.0, 0.0] and f[:, 2].tolist() == [0.0, 0.0, 0.0]
    assert f[0, 3:].tolist() == [2.5, 2.5, 2.5, 2.5]
    assert f[1, 3:].tolist() == [3.27, 2.5, 1.73, 2.5]
    assert f[2, 3:].tolist() == [3.27, 3.02, 1.31, 2.4]
    assert g.take_final_scores() == [32700, 49200, -5900, 24000]
    assert g.take_rank_by_player() == [1, 0, 3, 2]
    assert len(g.take_feature()) == 0  # taken


def _oracle_logs(oracle, n, policy, start):
    """n full game logs as text, played on the oracle arena (CPU) with the test policies."""
    from mortal_amd import mjai_log

    seeds = parity_util.default_seeds(n, start)
    arena = oracle.Arena(seeds, deal_algo=0, enable_quick_eval=True, version=3, keep_log=True)
    cyc = 0
    while arena.n_live > 0:
        rows = arena.poll()
        k = len(rows)
        obs, masks = arena.encode(0, k, want_obs=(policy == "greedy"))
        if policy == "greedy":
            act = parity_util.greedy_actions(masks, rows, cyc, obs[:, 919:922], 12345)
        else:
            act = oracle.random_actions(masks, rows, cyc)
        arena.commit(act)
        cyc += 1
    names = ["a", "b", "a", "c"]
    return [mjai_log.dump_json_log(names, seeds[g], arena.log(g)) for g in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("version,always_kan,augmented", [(3, True, False), (4, False, False), (2, True, True)])
def test_gameplay_loader_matches_reference_restatement(oracle, version, always_kan, augmented):
    """Every sample of every wanted player: obs and mask bit-exact, label / kyoku index / turn / shanten / gamma / done
    equal to the reference loader restated on the oracle (tests/dataset_ref.py)."""
    n_samples, n_kan = check_loader(oracle, version, always_kan, augmented, 6 if version == 4 else 12, 3)
    assert n_samples > 1500 and n_kan > 0


def check_loader(oracle, version, always_kan, augmented, n_greedy, n_random):
    """Shared by the GPU test above and the host-emulator test (tests/test_emu_device_code.py)."""
    import dataset_ref
    from libriichi.dataset import GameplayLoader
    from mortal_amd import mjai_log

    logs = [open(GOLDEN).read()] + _oracle_logs(oracle, n_greedy, "greedy", 4242) + _oracle_logs(oracle, n_random, "random", 99)
    loader = GameplayLoader(version, oracle=False, always_include_kan_select=always_kan, augmented=augmented,
                            player_names=["a", "mortal"])
    got = loader.load_logs(logs)
    assert len(got) == len(logs)
    n_samples = n_kan = 0
    for raw, games in zip(logs, got):
        events = [json.loads(l) for l in raw.splitlines()]
        if augmented:
            events = mjai_log.decode_events(mjai_log.encode_events(events, augmented=True))
            events = [json.loads(raw.splitlines()[0])] + events + [{"type": "end_game"}]
        wanted = [i for i, nm in enumerate(events[0]["names"]) if nm in ("a", "mortal")]
        assert [g.take_player_id() for g in games] == wanted
        for g in games:
            ref = dataset_ref.load_events_by_player(oracle, events, g.player_id, version, always_kan)
            assert g.player_name == ref["player_name"]
            assert g.actions == ref["actions"], (g.player_id, g.actions[:20], ref["actions"][:20])
            assert g.take_at_kyoku() == ref["at_kyoku"]
            assert g.take_dones() == ref["dones"]
            assert g.take_apply_gamma() == ref["apply_gamma"]
            assert g.take_at_turns() == ref["at_turns"]
            assert g.take_shantens() == ref["shantens"]
            obs, masks = g.take_obs(), g.take_masks()
            assert len(obs) == len(ref["obs"]) == len(masks)
            for k in range(len(obs)):
                assert (masks[k] == ref["masks"][k]).all(), (g.player_id, k)
                assert (obs[k].view(np.uint32) == ref["obs"][k].view(np.uint32)).all(), (g.player_id, k, ref["actions"][k])
            n_samples += len(obs)
            n_kan += sum(1 for a in ref["actions"] if a == 42)
            grp = g.take_grp()
            assert grp.take_feature().shape[1] == 7
    return n_samples, n_kan


@pytest.mark.gpu
@pytest.mark.parametrize("version", [1, 4])
def test_gameplay_loader_invisible_obs(oracle, version):
    """oracle=True: with trust_seed the wall is rebuilt from the game seed on the device and every invisible obs equals
    dataset/invisible.rs restated on the oracle; without it the opponents' planes still match and the wall planes hold
    the right number of tiles (the never-seen tiles are a random fill in the reference too)."""
    assert check_invisible(oracle, version, 5) > 500


@pytest.mark.gpu
def test_gameplay_loader_oracle_trust_seed_augmented(oracle):
    """The one flag combination round 2 refused: accepted by the reference (dataset/gameplay.rs:126-164)."""
    assert check_invisible(oracle, 4, 3, augmented=True) > 300


def check_invisible(oracle, version, n_logs, augmented=False):
    """augmented=True: GameplayLoader(oracle, trust_seed, augmented), which the reference accepts (gameplay.rs:126-164): the
    events are suit-swapped, the invisible wall is dealt from the seed as it was (invisible.rs:36-71)."""
    import dataset_ref
    from libriichi.dataset import GameplayLoader
    from mortal_amd import mjai_log

    logs = _oracle_logs(oracle, n_logs, "greedy", 31337)
    names = ["a", "c"]
    trusted = GameplayLoader(version, oracle=True, trust_seed=True, player_names=names, augmented=augmented).load_logs(logs)
    blind = GameplayLoader(version, oracle=True, trust_seed=False, player_names=names, augmented=augmented).load_logs(logs)
    opp_rows = 45 if version == 1 else 51
    n = 0
    for raw, gt, gb in zip(logs, trusted, blind):
        events = [json.loads(l) for l in raw.splitlines()]
        if augmented:
            head = events[0]
            events = [head] + mjai_log.decode_events(mjai_log.encode_events(events, augmented=True)) + [{"type": "end_game"}]
        for a, b in zip(gt, gb):
            ref = dataset_ref.load_invisible_by_player(oracle, events, a.player_id, version)
            inv_a, inv_b = a.take_invisible_obs(), b.take_invisible_obs()
            assert len(inv_a) == len(ref) == len(inv_b) == len(a.actions)
            for k in range(len(ref)):
                assert (inv_a[k].view(np.uint32) == ref[k].view(np.uint32)).all(), (a.player_id, k)
                assert (inv_b[k][:opp_rows] == ref[k][:opp_rows]).all()
                assert inv_b[k][opp_rows:].sum(axis=1)[::2].sum() == ref[k][opp_rows:].sum(axis=1)[::2].sum()
            n += len(ref)
    return n
