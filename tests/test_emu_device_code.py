"""The DEVICE code on the host: the unchanged HIP sources of mortal_amd/csrc, compiled against the fiber-based SIMT emulator
of tests/host/emu and driven through the same C-ABI and the same lock-step harness as the `-m gpu` parity tests
(tests/parity_util.py), against the oracle.  Runs without a GPU, so every kernel change is checked bit for bit before it
costs GPU time; the `-m gpu` tests remain the parity tests proper (real wavefronts, real memory model).

Small sizes: the emulator executes one work-item at a time (a v4 decision with a large SP state graph takes ~0.1 s)."""
import os
import shutil
import sys

import pytest

HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
if HOST not in sys.path:
    sys.path.insert(0, HOST)

import parity_util  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    import build_emu

    if not (os.path.exists(build_emu.CXX) or shutil.which("g++")):
        pytest.skip("no host C++ compiler")
    import emu_pool

    return emu_pool.make_pool_class()


def test_emu_lockstep_v3_full_hanchan_with_logs(oracle, emu):
    """step + row assignment + snapshot + encode<3> + event log kernels: two whole hanchan, every obs, every event."""
    st = parity_util.run_lockstep(oracle, 2, version=3, max_cycles=3000, obs_every=1, pool_cls=emu, compare_logs=True,
                                  deal_algo=1, verbose=False)
    assert st["scores_checked"] == 2 and st["obs_checked"] > 1500 and st["log_events_checked"] > 2500


def test_emu_lockstep_v4_sp_rows_greedy(oracle, emu):
    """mj_k_sp (expansion, level-0 scoring, evaluation) under the tenpai-seeking policy: f32 bit-exact SP rows."""
    st = parity_util.run_lockstep(oracle, 4, version=4, max_cycles=90, obs_every=1, pool_cls=emu, sp_rows_checked=True,
                                  policy="greedy", verbose=False)
    assert st["obs_checked"] > 300 and st["counters"]["sp_overflow"] == 0


def test_emu_lockstep_v4_one_workgroup_takes_every_row(oracle, emu, monkeypatch):
    """MJ_SP_GRID=1: a single persistent workgroup pops the whole row queue, so every row but the first starts from the state
    the previous one left behind (hash tags cleared, level bookkeeping, the queue tail handed to the wavefronts) — with the
    default grid the small test pools give each workgroup exactly one row and never exercise that."""
    monkeypatch.setenv("MJ_SP_GRID", "1")
    st = parity_util.run_lockstep(oracle, 4, version=4, max_cycles=60, obs_every=1, pool_cls=emu, sp_rows_checked=True,
                                  policy="greedy", verbose=False)
    assert st["obs_checked"] > 200 and st["counters"]["sp_overflow"] == 0


def test_emu_lockstep_older_obs_versions_and_guard(oracle, emu):
    for version in (1, 2):
        st = parity_util.run_lockstep(oracle, 2, version=version, max_cycles=150, obs_every=1, pool_cls=emu, policy="greedy",
                                      verbose=False)
        assert st["obs_checked"] > 250
    st = parity_util.run_lockstep(oracle, 4, version=3, max_cycles=700, obs_every=25, pool_cls=emu, policy="greedy",
                                  guard=True, quick_eval=False, verbose=False)
    assert st["cycles"] == 700


def test_emu_lockstep_refill(oracle, emu):
    """mj_k_refill (the benchmark's steady-state mode): slots restart on nonce + stride; >= 2 hanchan per slot."""
    st = parity_util.run_lockstep(oracle, 2, version=3, max_cycles=20000, obs_every=9, pool_cls=emu, refill=8, min_games=2,
                                  deal_algo=1, verbose=False)
    assert st["generations"][0] >= 2 and st["games_checked"] >= 4


def test_emu_lockstep_staggered_start(oracle, emu):
    """The benchmark's protocol (set_refill + set_start_stagger) in lock-step with the oracle: slots idle until their own cycle,
    start on nonce + stride, then restart like refilled tables; obs v4 with the SP rows on the first, v3 + greedy on the second."""
    st = parity_util.run_lockstep(oracle, 3, version=4, max_cycles=20000, obs_every=40, sp_rows_checked=True, pool_cls=emu, refill=8,
                                  stagger=90, min_games=1, deal_algo=1, verbose=False)
    assert st["generations"][0] >= 2 and st["games_checked"] >= 3 and st["counters"]["sp_overflow"] == 0
    st = parity_util.run_lockstep(oracle, 4, version=3, max_cycles=20000, obs_every=9, pool_cls=emu, refill=8, stagger=300,
                                  min_games=2, policy="greedy", verbose=False)
    assert st["generations"][0] >= 3 and st["games_checked"] >= 8


def test_emu_invisible_obs(oracle, emu):
    st = parity_util.run_lockstep(oracle, 2, version=3, max_cycles=200, obs_every=2, compare_obs=False, pool_cls=emu,
                                  policy="greedy", oracle_obs=True, verbose=False)
    assert st["oracle_obs_checked"] > 150


def test_emu_device_engine_sampling_logs_and_stat(oracle, emu, tmp_path):
    """The arena end to end on the emulated kernels with a `react_batch_device` engine that explores (Boltzmann-epsilon /
    top-p on the device path) and returns (actions, q_values, is_greedy): logs are written with the reference's per-decision
    `meta` objects (is_greedy False on explored decisions), `Stat.from_dir` reads them back, never an illegal action."""
    import gzip
    import json

    import torch
    from libriichi.arena import OneVsThree
    from libriichi.stat import Stat

    from mortal_amd import arena as A
    from mortal_amd.policy import DeviceEngine, PolicyNet

    def engine(seed, name, eps):
        torch.manual_seed(seed)
        return DeviceEngine(PolicyNet(version=3, conv_channels=16, num_blocks=1), 3, "cpu", name=name, enable_amp=False,
                            boltzmann_epsilon=eps, boltzmann_temp=1.0, top_p=0.9, return_meta=True, seed=seed)

    old = A.BatchRunner.pool_cls
    A.BatchRunner.pool_cls = emu
    try:
        d = str(tmp_path / "logs")
        got = OneVsThree(disable_progress_bar=True, log_dir=d).py_vs_py(challenger=engine(1, "challenger", 0.3),
                                                                        champion=engine(2, "champion", 0.0),
                                                                        seed_start=(10000, 0xD5DFAA4CEF265CD7), seed_count=1)
    finally:
        A.BatchRunner.pool_cls = old
    assert sum(got) == 4
    st = Stat.from_dir(d, "challenger", True)
    assert st.game == 4 and [st.rank_1, st.rank_2, st.rank_3, st.rank_4] == got
    n_meta = n_explored = 0
    for f in sorted(os.listdir(d)):
        for line in gzip.open(os.path.join(d, f), "rt"):
            ev = json.loads(line)
            meta = ev.get("meta")
            if meta is None:
                continue
            n_meta += 1
            assert len(meta["q_values"]) == bin(meta["mask_bits"]).count("1") and "shanten" in meta and "batch_size" in meta
            n_explored += not meta["is_greedy"]
    assert n_meta > 500 and 0 < n_explored < n_meta


def test_emu_dataset_loader(oracle, emu):
    """libriichi.dataset.GameplayLoader on the emulated kernels (mj_k_replay + labels + encode, suit augmentation, and the
    invisible obs with the wall rebuilt from the seed — the logs come from an oracle arena dealing with the OTHER rand
    generation than the pool's default, so the replay's shuffle fallback runs too): the GPU tests of tests/test_dataset.py at
    a reduced size."""
    import test_dataset as TD
    from mortal_amd.dataset import GameplayLoader

    old = GameplayLoader.pool_cls
    GameplayLoader.pool_cls = emu
    try:
        n_samples, _ = TD.check_loader(oracle, 3, True, False, 1, 1)
        assert n_samples > 150
        n_samples, _ = TD.check_loader(oracle, 2, True, True, 1, 0)
        assert n_samples > 80
        assert TD.check_invisible(oracle, 4, 1) > 60
        assert TD.check_invisible(oracle, 3, 1, augmented=True) > 60  # oracle + trust_seed + augmented (gameplay.rs:126-164)
    finally:
        GameplayLoader.pool_cls = old


def test_emu_reference_state_scenarios_and_bot(oracle, emu):
    """tests/test_gpu_state.py on the emulated kernels: the reference's own state/test.rs scenarios applied to the device event
    handlers (mj_table_apply_event), single-table encode_obs of all four versions, validate_reaction, libriichi.mjai.Bot and
    the reference's encode-obs benchmark kyoku for every seat."""
    import test_gpu_state as G

    from mortal_amd.state import PlayerState

    old = PlayerState.pool_cls
    PlayerState.pool_cls = emu
    try:
        for name in G.EVENT_DRIVEN:
            G.test_reference_state_scenario_on_device(name)
        G.test_reference_waits_vectors_on_device()
        G.test_reference_can_chi_vectors_on_device()
        G.test_device_player_state_obs_matches_oracle(oracle)
        G.test_device_player_state_validate_reaction(oracle)
        G.test_bot_replays_a_game_like_the_oracle_agent(oracle)
        for pid in range(4):
            G.test_reference_bench_kyoku_obs_device_vs_oracle(oracle, pid)
    finally:
        PlayerState.pool_cls = old


def test_emu_reference_kats_and_generated_hands(oracle, emu):
    """tests/test_gpu_kats.py on the emulated kernels: the reference's shanten / ankan / agari / point KATs through
    `mj_algo_query`, and generated complete hands with random melds device vs oracle (reduced count)."""
    import test_gpu_kats as K

    lib = emu._L
    st = K.check_reference_kats(oracle, lib)
    assert st["shanten"] == 19 and st["agari"] == 25 and st["open"] >= 5
    assert K.check_point_sweep(lib) > 250
    assert K.check_deal_divmod(lib, per_n=100) > 20_000
    st = K.check_generated_hands(oracle, 4000, 7, lib)
    assert st["open"] > 1000 and st["kans"] > 500 and st["yakuman"] > 30 and st["none"] > 30, st


def test_emu_staggered_start(oracle, emu):
    """mj_pool_set_start_stagger (the benchmark's steady-state mode): every table is parked, enters play at its own cycle
    hash(t) % S through the refill path, and from then on behaves like any refilled table (no errors, games finish)."""
    import numpy as np
    import torch

    n, S = 24, 40
    pool = emu(n, version=3, max_rows=4 * n)
    pool.reset(parity_util.default_seeds(n), game_ids=np.arange(n), n_games_total=n)
    pool.set_refill(n)
    pool.set_start_stagger(S)
    masks = torch.zeros((4 * n, 46), dtype=torch.bool)
    obs = torch.zeros((4 * n, 934, 34), dtype=torch.float32)
    act = torch.zeros(4 * n, dtype=torch.int32)
    starts = [((t * 2654435761) % (1 << 32) >> 8) % S for t in range(n)]
    live_prev, a_prev = 0, None
    for c in range(S + 400):
        k, _ = pool.step(a_prev, None)
        steps = pool.counters()["steps"]
        live = steps - live_prev
        live_prev = steps
        assert live == sum(1 for s in starts if s <= c), (c, live)  # the refill of cycle s runs ahead of that cycle's step: in play from cycle s on
        pool.encode(0, obs, masks)
        pool.random_policy(0, masks, 7, c, act)
        a_prev = act[:k]
        if c > S:
            assert k > 0
    assert pool.first_error()[0] == 0


def test_emu_tag_epoch_wrap_in_a_variant_build():
    """The hash tags of mj_k_sp carry the row's epoch instead of being cleared between rows (round 5); when a workgroup's epoch
    counter wraps -- once in 2 M rows with a state graph, about half an hour of the benchmark's launches -- the table is wiped and the
    count starts over.  No regular test gets near that, so a variant build of the emulator library with the wrap after THREE rows
    (-DSP_EPOCH_WRAP=3) runs the lock-step test in which one workgroup takes every row (MJ_SP_GRID=1): dozens of wraps, SP rows f32
    bit-exact against the oracle, no overflow.  A subprocess: the emulator library of this process is built without the flag."""
    import subprocess

    env = dict(os.environ, EMU_EXTRA_FLAGS="-DSP_EPOCH_WRAP=3")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_emu_device_code.py"), "-x", "-q", "-k",
                          "one_workgroup_takes_every_row", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_emu_lockstep_v4_every_large_row_promoted_to_the_wide_kernel(oracle, emu, monkeypatch):
    """Small-pool schedule of round 6 (mj_sp.hip "promotion"): with MJ_SP_WIDE=1 and level thresholds of 1 every row with two or more
    levels is parked by its 256-thread workgroup after the first expansion and finished by mj_k_sp_wide (1,024 threads) from the hand-off
    block of its work area — remaining expansion levels, evaluation and row output.  Same f32 bits as the oracle; the emulator runs the
    launches one after the other, so the wide kernel takes the rows as the sweep launch does on the GPU."""
    monkeypatch.setenv("MJ_SP_WIDE", "1")
    monkeypatch.setenv("MJ_SP_PROMO_MIN1", "1")
    monkeypatch.setenv("MJ_SP_PROMO_MIN2", "1")
    st = parity_util.run_lockstep(oracle, 4, version=4, max_cycles=70, obs_every=1, pool_cls=emu, sp_rows_checked=True,
                                  policy="greedy", verbose=False)
    assert st["obs_checked"] > 250 and st["counters"]["sp_overflow"] == 0
    assert st["sp_schedule"]["hybrid_launches"] > 50 and st["sp_schedule"]["rows_swept"] > 100, st["sp_schedule"]
    # one workgroup chaining rows: after a promotion it continues in a spare work area (its tag epoch, its cache)
    monkeypatch.setenv("MJ_SP_GRID", "1")
    st = parity_util.run_lockstep(oracle, 4, version=4, max_cycles=50, obs_every=1, pool_cls=emu, sp_rows_checked=True,
                                  policy="random", verbose=False)
    assert st["obs_checked"] > 150 and st["counters"]["sp_overflow"] == 0 and st["sp_schedule"]["rows_swept"] > 30, st["sp_schedule"]
    # the wide kernel as a worker of the row queue (on the GPU it takes rows from the queue whenever no promoted row waits): every row
    # with a state graph from set-up to output by 1,024 threads
    monkeypatch.delenv("MJ_SP_GRID")
    monkeypatch.setenv("MJ_SP_WIDE_ALL_ROWS", "1")
    st = parity_util.run_lockstep(oracle, 4, version=4, max_cycles=50, obs_every=1, pool_cls=emu, sp_rows_checked=True,
                                  policy="greedy", verbose=False)
    assert st["obs_checked"] > 150 and st["counters"]["sp_overflow"] == 0 and st["sp_schedule"]["rows_swept"] == 0, st["sp_schedule"]


def test_emu_sp_schedule_api(emu):
    """include/mortal_amd.h mj_pool_set_sp_schedule / mj_sp_schedule_stats: mode 0 before the first obs-v4 encode = no spare work areas, every
    launch runs mj_k_sp alone; asking for the schedule afterwards is refused (the areas are sized at the first encode); thresholds may change."""
    import numpy as np
    import torch

    from mortal_amd._lib import MortalAmdError

    pool = emu(4, version=4)
    pool.reset(parity_util.default_seeds(4), game_ids=np.arange(4), n_games_total=4)
    pool.set_sp_schedule(mode=0)
    act = None
    for i in range(12):
        n, _ = pool.step(act, None)
        obs, masks = pool.encode(0)
        act = pool.random_policy(0, masks, 1, i)
    assert pool.sp_schedule_stats() == {"hybrid_launches": 0, "rows_promoted": 0, "rows_swept": 0, "wide_gave_up": 0}
    with pytest.raises(MortalAmdError):
        pool.set_sp_schedule(mode=1)
    pool.set_sp_schedule(min_level1=900, min_level2=300)  # (allowed: only mode needs the areas)
    pool.close()
