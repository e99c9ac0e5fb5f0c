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
    """BASELINE configs[1] size: 4096 tables, masks + row lists every cycle, the whole v3 obs of every decision on a
    sampled cycle set (every 89th cycle, ~4,100 rows each), final scores of every game."""
    st = parity_util.run_lockstep(oracle, 4096, version=3, max_cycles=3000, obs_cycles=set(range(3, 3000, 89)), threads=16)
    assert st["scores_checked"] == 4096 and st["obs_checked"] > 50000
    assert st["counters"]["steps"] == st["oracle_steps"]


def test_lockstep_4096_tables_v4_obs_with_sp(oracle):
    """The same pool size with obs v4: all 1012 rows incl. the SP block on sampled cycles of the first kyoku (the most
    SP-heavy phase: every table at 17 draws left) and later ones."""
    st = parity_util.run_lockstep(oracle, 4096, version=4, max_cycles=420, obs_cycles={2, 37, 111, 222, 333, 419},
                                  sp_rows_checked=True, threads=16)
    assert st["obs_checked"] > 15000 and st["counters"]["sp_overflow"] == 0
    # a pool of this size runs under the small-pool schedule (round 6): rows with a large state graph were parked by mj_k_sp_promo and
    # finished by mj_k_sp_wide -- with the default thresholds, in every launch of the SP-heavy first turns
    sc = st["sp_schedule"]
    print("small-pool schedule:", sc)
    # (the two kernels have run side by side on every box so far; where they do not, the wide workgroups give up, the sweep launch
    # finishes the parked rows and the schedule switches itself off -- parity holds either way, so only the first launch is demanded)
    assert sc["hybrid_launches"] >= 1 and (sc["wide_gave_up"] > 0 or (sc["hybrid_launches"] >= 420 and sc["rows_promoted"] > 1000)), sc


def test_lockstep_small_pool_schedule_parks_every_row_it_can(oracle, monkeypatch):
    """The small-pool schedule with thresholds of a few states (mj_sp.hip "promotion"; MJ_SP_WIDE=1, level sizes 24 / 6): nearly every row
    with two or more levels is parked after its first or second expansion and finished by mj_k_sp_wide from the hand-off block, while the
    wide workgroups take whole rows of the queue in between -- REAL concurrency of the two kernels on two streams (the emulator runs them
    one after the other), agent-scope release / acquire across CUs and XCDs, spare work areas with their own tag epochs, two promotion
    queues.  Rows, masks and the whole v4 obs against the oracle under refill + stagger (every phase of a hanchan in the queue)."""
    monkeypatch.setenv("MJ_SP_WIDE", "1")
    monkeypatch.setenv("MJ_SP_PROMO_MIN1", "24")
    monkeypatch.setenv("MJ_SP_PROMO_MIN2", "6")
    monkeypatch.setenv("MJ_SP_WIDE_GRID", "48")
    st = parity_util.run_lockstep(oracle, 512, version=4, max_cycles=600, obs_every=7, sp_rows_checked=True, refill=128, stagger=300,
                                  min_games=99, deal_algo=1, threads=16)
    sc = st["sp_schedule"]
    assert st["obs_checked"] > 20000 and st["counters"]["sp_overflow"] == 0
    print("small-pool schedule, every eligible row parked:", sc)
    assert sc["rows_promoted"] + sc["rows_swept"] > 20000, sc  # (swept = finished by the sweep launch: the kernels did not overlap on this box)


def test_lockstep_rand09_deal(oracle):
    """deal_algo = MJ_DEAL_RAND09 (the shuffle of rand 0.9.1, the generation the reference's Cargo.lock pins and the
    pool's default): device deal == oracle deal (tests/test_oracle_deal.py pins the oracle against a second
    implementation), whole hanchan with event logs and obs."""
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=3000, obs_every=3, deal_algo=1, compare_logs=True)
    assert st["scores_checked"] == 256 and st["log_events_checked"] > 100000
    st = parity_util.run_lockstep(oracle, 32, version=4, max_cycles=3000, obs_every=9, deal_algo=1, policy="greedy",
                                  sp_rows_checked=True)
    assert st["scores_checked"] == 32 and st["counters"]["sp_overflow"] == 0


def test_lockstep_refill_matches_benchmark_mode(oracle):
    """What bench.py times: mj_pool_set_refill — a finished table restarts on (nonce + stride, key) with game id + N.
    The oracle arena restarts its slots the same way; rows, masks, v4 obs incl. SP rows, step counter and the final
    scores of every finished hanchan (>= 2 per slot, i.e. tables in their 2nd / 3rd game like the timed window)."""
    st = parity_util.run_lockstep(oracle, 48, version=4, max_cycles=20000, obs_every=6, sp_rows_checked=True, refill=12,
                                  min_games=2, deal_algo=1, threads=8)
    assert st["generations"][0] >= 2 and st["games_checked"] >= 96 and st["counters"]["sp_overflow"] == 0
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=20000, obs_every=11, refill=64, min_games=3,
                                  policy="greedy", threads=8)
    assert st["generations"][0] >= 3 and st["games_checked"] >= 768


def test_lockstep_staggered_start_matches_benchmark_protocol(oracle):
    """The protocol the headline is TIMED under (bench.py: set_refill + set_start_stagger): every table parked, table t entering
    play at cycle hash(t) % S through the refill path, then restarting like any refilled table.  The oracle slots idle and start
    the same way; rows, masks, obs v4 incl. the SP rows, the step counter and the scores of >= 2 played hanchan per slot."""
    st = parity_util.run_lockstep(oracle, 48, version=4, max_cycles=20000, obs_every=6, sp_rows_checked=True, refill=12,
                                  stagger=700, min_games=2, deal_algo=1, threads=8)
    assert st["generations"][0] >= 3 and st["games_checked"] >= 96 and st["counters"]["sp_overflow"] == 0
    st = parity_util.run_lockstep(oracle, 256, version=3, max_cycles=20000, obs_every=11, refill=64, stagger=1500, min_games=2,
                                  policy="greedy", threads=8)
    assert st["generations"][0] >= 3 and st["games_checked"] >= 512


def test_lockstep_quick_eval_disabled(oracle):
    """enable_quick_eval = False (mortal.rs:210-250): single-candidate discards get a row, every ankan/kakan decision
    gets a kan-select row."""
    st = parity_util.run_lockstep(oracle, 128, version=3, max_cycles=3000, obs_every=2, quick_eval=False, policy="greedy")
    assert st["scores_checked"] == 128 and st["counters"]["quick"] == 0


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


def test_lockstep_tsumogiri_reference_seeds(oracle):
    """arena/game.rs:323-371 (the reference's only whole-arena test): four Tsumogiri agents on seeds (1009, 0) and
    (1021, 0); the device pool follows the oracle event for event through all twelve kyoku (E1 .. W4)."""
    st = parity_util.run_lockstep(oracle, 2, version=3, seeds=[(1009, 0), (1021, 0)], policy="tsumogiri", compare_logs=True,
                                  max_cycles=3000)
    assert st["done_gpu"] == 2 and st["done_oracle"] == 2 and st["scores_checked"] == 2
    assert st["log_events_checked"] == 2 * 1716


@pytest.mark.parametrize("version", [1, 2])
def test_lockstep_obs_v1_v2(oracle, version):
    """The two older obs layouts (938 / 942 planes: thermometer integers, no decay rows / RBF rows) under the greedy
    policy, every obs compared."""
    st = parity_util.run_lockstep(oracle, 128, version=version, max_cycles=400, obs_every=1, policy="greedy")
    assert st["obs_checked"] > 40000


def test_full_size_pool_against_the_oracle(oracle):
    """The headline pool size against the ORACLE (the 65,536-table pool had only been compared with a smaller HIP
    pool): 65,536 tables, obs v3, uniform-random legal policy, the first 48 cycles (~3.2 M decisions) — row lists and 46-wide
    masks of every decision of every cycle, the whole obs tensor on sampled cycles, the step counter.  The oracle arena holds
    all 65,536 games and encodes on 16 threads."""
    st = parity_util.run_lockstep(oracle, 65536, version=3, max_cycles=48, obs_every=12, threads=16, obs_slice=8192, verbose=True)
    assert st["rows"] > 3_000_000 and st["obs_checked"] > 250_000 and st["cycles"] == 48
    assert st["counters"]["steps"] == st["oracle_steps"] == 48 * 65536


def test_full_size_pool_v4_sp_staggered_against_the_oracle(oracle):
    """The HEADLINE configuration and protocol against the oracle (at 65,536 tables the SP rows had only been compared
    HIP-vs-HIP from a cold start): 65,536 tables, obs v4, uniform-random legal policy, refill + staggered first starts
    (mj_pool_set_start_stagger, 200 cycles) so that the row queue of mj_k_sp holds every cost class at once and every persistent
    workgroup takes MANY graph rows per launch (tag clearing, child-cache epochs, the one-row-per-wavefront tail) — row lists and
    46-wide masks of every decision of 240 cycles, the WHOLE v4 obs including rows 889..1011 (f32 bit for bit, NaN-poisoned
    buffers) on two sampled cycles for two 4,096-row slices each, the step counter, no SP overflow.
    Reference: agent/mortal.rs:252-287, state/obs_repr.rs:564-624."""
    st = parity_util.run_lockstep(oracle, 65536, version=4, max_cycles=240, obs_cycles={150, 239}, sp_rows_checked=True, refill=65536 // 4,
                                  stagger=200, min_games=99, deal_algo=1, threads=16, obs_slice=4096, obs_slice_max=2, verbose=True)
    assert st["cycles"] == 240 and st["rows"] > 9_000_000 and st["obs_checked"] >= 4 * 4096
    assert st["counters"]["sp_overflow"] == 0 and st["counters"]["steps"] == st["oracle_steps"]


def test_benchmark_protocol_16384_tables_v4_sp_late_game_against_the_oracle(oracle):
    """The benchmark's REAL protocol at a size where the persistent workgroups of mj_k_sp chain many rows (the 65,536-table
    comparison staggers over 200 cycles and stops at cycle 240, so it only ever sees tables in the first kyoku of their first hanchan;
    the late-game mix had been oracle-checked at 48 tables): 16,384 tables, obs v4, uniform-random legal policy, refill, first starts
    staggered over 3,072 cycles exactly like `bench.py` -- through that pre-roll the oracle follows with masks and row lists only --
    then 64 more cycles in which the row queue mixes East and South rounds, all-last, riichi-heavy late turns and second-generation
    (refilled) tables: row lists and 46-wide masks of every decision of all 3,136 cycles, the WHOLE v4 obs including rows 889..1011
    (f32 bit for bit, NaN-poisoned buffers) on two 4,096-row slices of two cycles after the pre-roll, the step counter, no SP overflow,
    and tables in their second hanchan present.  Reference: agent/mortal.rs:252-287, state/obs_repr.rs:564-624, arena/game.rs:286-304."""
    pre = 3072
    st = parity_util.run_lockstep(oracle, 16384, version=4, max_cycles=pre + 64, obs_cycles={pre + 21, pre + 63}, sp_rows_checked=True,
                                  refill=16384 // 4, stagger=pre, min_games=99, deal_algo=1, threads=16, obs_slice=4096, obs_slice_max=2,
                                  verbose=True)
    assert st["cycles"] == pre + 64 and st["obs_checked"] >= 4 * 4096
    assert st["generations"][1] >= 2  # (generation 1 = the staggered first start) some slots are in their second played hanchan
    assert st["counters"]["sp_overflow"] == 0 and st["counters"]["steps"] == st["oracle_steps"]


@pytest.mark.parametrize("version,cycles", [(3, 48), (4, 14)])
def test_full_size_pool_is_size_independent(version, cycles):
    """BASELINE's 65,536-table configuration: tables are independent, so the first 1,024 tables of the big pool must
    produce exactly the rows / masks / obs of a 1,024-table pool on the same seeds and policy, cycle after cycle; plus
    whole-pool invariants (no table in error, every mask row has a legal action, every live table steps every cycle)."""
    import numpy as np
    import torch

    from mortal_amd.pool import TablePool

    big_n, small_n = 65536, 1024
    seeds = parity_util.default_seeds(big_n)
    big = TablePool(big_n, version=version, max_rows=2 * big_n)
    small = TablePool(small_n, version=version)
    big.reset(seeds, game_ids=np.arange(big_n), n_games_total=big_n)
    small.reset(seeds[:small_n], game_ids=np.arange(small_n), n_games_total=small_n)
    obs = torch.empty((2 * big_n, big.C, 34), dtype=torch.float32, device="cuda")
    masks = torch.empty((2 * big_n, 46), dtype=torch.bool, device="cuda")
    act_b = act_s = None
    for cyc in range(cycles):
        nb, _ = big.step(act_b, None)
        ns, _ = small.step(act_s, None)
        ob, mb = big.encode(0, obs, masks)
        os_, ms = small.encode(0)
        rows_b, rows_s = big.rows(0), small.rows(0)
        sel = np.flatnonzero(rows_b[:, 0] < small_n)  # rows are ordered by table: the small pool's tables come first
        assert len(sel) == ns and (rows_b[sel] == rows_s).all()
        idx = torch.as_tensor(sel, device="cuda")
        assert torch.equal(mb[idx], ms)
        assert torch.equal(ob[idx].view(torch.int32), os_.view(torch.int32))
        assert bool(mb.any(dim=1).all())
        act_b = big.random_policy(0, mb, 0x9E3779B97F4A7C15, cyc).clone()
        act_s = small.random_policy(0, ms, 0x9E3779B97F4A7C15, cyc).clone()
    assert big.first_error()[0] == 0 and small.first_error()[0] == 0
    c = big.counters()
    assert c["steps"] == cycles * big_n and c["games"] == 0 and c["sp_overflow"] == 0
    big.close()
    small.close()
