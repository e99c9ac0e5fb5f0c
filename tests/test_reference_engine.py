"""BASELINE configs[0] as written: the reference's own `mortal/engine.py` (MortalEngine) and `mortal/model.py` (Brain, DQN),
imported UNCHANGED from /root/reference with this repository's `libriichi` package on the path, playing
`libriichi.arena.OneVsThree.py_vs_py` — the call `mortal/one_vs_three.py:77-97` makes — and checked against the
reference-shaped oracle loop driven by the same two networks through the reference's list-of-ndarray contract.

Runs on CPU: the arena's device kernels execute on the host SIMT emulator (tests/host/emu), the networks on torch CPU.
Skipped where /root/reference does not exist (the GPU box)."""
import os
import sys

import numpy as np
import pytest

REF = os.environ.get("MORTAL_REF_DIR", "/root/reference/mortal")
REAL = os.environ.get("MORTAL_AMD_CFG0_REAL") == "1"  # tools/r06_cfg0_gpu.sh: the real libmortal_amd.so and the networks on cuda:0
DEVICE = "cuda:0" if REAL else "cpu"
KEY = 0xD5DFAA4CEF265CD7
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "engine.py")), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref_modules():
    for p in (ROOT, os.path.join(ROOT, "compat"), HOST, REF):
        if p not in sys.path:
            sys.path.append(p)  # the repository's `libriichi` must win over anything in the reference tree
    import engine as ref_engine  # noqa: E402  (mortal/engine.py)
    import model as ref_model  # noqa: E402  (mortal/model.py: imports libriichi.consts)

    import libriichi

    assert os.path.dirname(libriichi.__file__).startswith(ROOT)
    return ref_engine, ref_model


def _engine(ref_engine, ref_model, version, seed, name, **kw):
    import torch

    torch.manual_seed(seed)
    brain = ref_model.Brain(version=version, conv_channels=16, num_blocks=1).to(DEVICE)  # "random-init tiny model"
    dqn = ref_model.DQN(version=version).to(DEVICE)
    return ref_engine.MortalEngine(brain, dqn, is_oracle=False, version=version, device=torch.device(DEVICE), name=name,
                                   enable_amp=False, enable_quick_eval=True, enable_rule_based_agari_guard=False, **kw)


def _oracle_rankings(oracle, chal, cham, seed_start, seed_count, version, deal_algo):
    """BatchGame::run (arena/game.rs:286-304) on the oracle; engines get the reference's contract: lists of per-row arrays."""
    n = seed_count * 4
    seeds = [(seed_start[0] + g // 4, seed_start[1]) for g in range(n)]
    arena = oracle.Arena(seeds, deal_algo=deal_algo, enable_quick_eval=True, version=version, keep_log=False)
    while arena.n_live > 0:
        rows = arena.poll()
        k = len(rows)
        obs, masks = arena.encode(0, k, want_obs=True)
        act = np.full(k, 45, dtype=np.int32)
        if k:
            is_chal = rows[:, 1] == rows[:, 0] % 4
            for eng, sel in ((chal, is_chal), (cham, ~is_chal)):
                idx = np.flatnonzero(sel)
                if len(idx):
                    a, _q, _m, _g = eng.react_batch([obs[i] for i in idx], [masks[i].astype(bool) for i in idx], None)
                    act[idx] = a
        arena.commit(act)
    rankings = [0, 0, 0, 0]
    for g in range(n):
        sc = arena.result(g)[0]
        order = sorted(range(4), key=lambda i: -int(sc[i]))
        rankings[order.index(g % 4)] += 1
    return rankings


@pytest.mark.parametrize("version", [4, 2])
def test_reference_engine_plays_one_vs_three(oracle, ref_modules, version):
    from libriichi.arena import OneVsThree

    from mortal_amd import arena as A

    ref_engine, ref_model = ref_modules
    chal = _engine(ref_engine, ref_model, version, 1, "challenger")
    cham = _engine(ref_engine, ref_model, version, 2, "champion")
    old = A.BatchRunner.pool_cls
    if not REAL:  # (REAL: the product's own TablePool over libmortal_amd.so)
        import emu_pool

        A.BatchRunner.pool_cls = emu_pool.make_pool_class()
    try:
        env = OneVsThree(disable_progress_bar=True)
        got = env.py_vs_py(challenger=chal, champion=cham, seed_start=(10000, KEY), seed_count=1)
    finally:
        A.BatchRunner.pool_cls = old
    from mortal_amd.pool import default_deal_algo

    want = _oracle_rankings(oracle, chal, cham, (10000, KEY), 1, version, default_deal_algo())
    assert sum(got) == 4 and got == want
    print(f"reference engine.py + model.py, py_vs_py v{version} on {'libmortal_amd.so, ' + DEVICE if REAL else 'the host emulator'}: rankings {got} == oracle loop")
