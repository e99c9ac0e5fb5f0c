"""-m gpu: the HIP table pool vs the oracle, bit for bit, through the C-ABI (BASELINE configs[1]: random-action
policy, env-step kernels; state transitions, masks, obs and scores must be identical)."""
import pytest

import parity_util

pytestmark = pytest.mark.gpu


def test_lockstep_v3_256_tables(oracle):
    """256 tables, full hanchan, every cycle: row lists, 46-wide masks and the whole v3 obs (934x34 f32) bit-exact."""
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=3000, obs_every=1)
    assert st["scores_checked"] == 256 and st["obs_checked"] > 100000


def test_lockstep_v4_full_obs_with_sp(oracle):
    """v4: all 1012 rows incl. the SP block (rows 889..1011, f32 bit-exact), sampled every 7th cycle; masks every cycle."""
    st = parity_util.run_lockstep(oracle, 64, version=4, max_cycles=3000, obs_every=7, sp_rows_checked=True)
    assert st["scores_checked"] == 64 and st["counters"]["sp_overflow"] == 0


def test_lockstep_4096_tables_masks_scores(oracle):
    """BASELINE configs[1] size: 4096 tables, masks + row lists every cycle, final scores of every game."""
    st = parity_util.run_lockstep(oracle, 4096, version=3, max_cycles=3000, compare_obs=False)
    assert st["scores_checked"] == 4096
    assert st["counters"]["steps"] == st["oracle_steps"]


def test_lockstep_greedy_policy_v3(oracle):
    """A shanten-greedy policy (always agari, mostly riichi, calls) drives the games through tenpai / riichi / ron /
    furiten / kan paths that uniform-random play rarely reaches; v3 obs compared every cycle."""
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=3000, obs_every=1, policy="greedy")
    assert st["scores_checked"] == 256


def test_lockstep_greedy_policy_v4_sp(oracle):
    """Same policy with obs v4: hands sit at 0..3 shanten most of the time, so the SP block is exercised with real
    probability tables (f32 bit-exact)."""
    st = parity_util.run_lockstep(oracle, 32, version=4, max_cycles=3000, obs_every=4, policy="greedy",
                                  sp_rows_checked=True)
    assert st["scores_checked"] == 32 and st["counters"]["sp_overflow"] == 0


def test_lockstep_rule_based_agari_guard(oracle):
    """enable_rule_based_agari_guard (agent/mortal.rs:319-336 + agent_helper.rs:251-368): both sides get the same
    synthetic q-values (16 levels => ties, -inf on illegal actions); an agari the rule engine rejects must turn into the
    same alternative action on both sides.  The greedy policy always answers agari, so all-last 4th-place hands hit it."""
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=4000, obs_every=16, policy="greedy", guard=True)
    assert st["scores_checked"] == 256
    assert st["guard_hits"] > 0


@pytest.mark.parametrize("version", [1, 3])
def test_lockstep_invisible_obs(oracle, version):
    """a14: BoardState::encode_oracle_obs (board.rs:679-782) — other seats' hands / shanten / waits / furiten, the live
    wall, rinshan, dora and ura indicators — bit-exact for every decision row, v1 (211 planes) and v2+ (217 planes).
    The greedy policy reaches kans (rinshan draws shift the wall window) and furiten."""
    st = parity_util.run_lockstep(oracle, 128, version=version, max_cycles=1500, obs_every=3,
                                  compare_obs=False, policy="random" if version == 1 else "greedy", oracle_obs=True)
    assert st["oracle_obs_checked"] > 20000


def test_lockstep_event_logs(oracle):
    """the device event log, decoded to mjai JSON, equals the oracle's game log (the reference's
    BoardState log, pinned by the golden game) event for event — start_kyoku haipai, draws, calls with aka-aware consumed
    tiles, kan doras, riichi acceptance, hora deltas + ura markers, ryukyoku deltas, end_kyoku."""
    st = parity_util.run_lockstep(oracle, 128, version=3, max_cycles=4000, compare_obs=False, policy="greedy",
                                  compare_logs=True)
    assert st["scores_checked"] == 128 and st["log_events_checked"] > 100000
    st = parity_util.run_lockstep(oracle, 64, version=3, max_cycles=4000, compare_obs=False, policy="random",
                                  compare_logs=True, seeds=parity_util.default_seeds(64, 777))
    assert st["scores_checked"] == 64
