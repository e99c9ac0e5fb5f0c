"""world_size-2 gloo test of the N>1 path: shard ranges, seed planning, the gather of episode returns and the
rank-histogram reduction.  Runs on CPU; the per-rank 'arena' is the oracle."""
import os

import numpy as np
import torch.multiprocessing as mp

KEY = 0xD5DFAA4CEF265CD7


def _play(seeds):
    """Tsumogiri-like fixed policy on the oracle: always the lowest legal action id (deterministic, fast)."""
    import oracle_lib as O

    arena = O.Arena(seeds, version=3, keep_log=False)
    while arena.n_live > 0:
        rows = arena.poll()
        _, masks = arena.encode(0, len(rows), want_obs=False)
        arena.commit(masks.argmax(axis=1).astype(np.int32) if len(rows) else np.zeros(0, np.int32))
    return np.array([arena.result(g)[0] for g in range(len(seeds))], dtype=np.int32)


def _worker(rank, world, port, n_games, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from mortal_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g0, g1 = sharding.shard_range(n_games, rank, world)
    scores = _play(sharding.seeds_for((10000, KEY), g0, g1))
    hist = [0, 0, 0, 0]
    for i, g in enumerate(range(g0, g1)):
        order = sorted(range(4), key=lambda s: -int(scores[i][s]))
        hist[order.index(g % 4)] += 1
    total = sharding.allreduce_rank_histogram(hist)
    allsc = sharding.gather_returns(scores, n_games)
    if rank == 0:
        q.put((total, allsc))
    dist.destroy_process_group()


def test_shard_ranges_cover_and_align():
    from mortal_amd.sharding import shard_range

    for n, w in [(8, 2), (12, 8), (524288, 8), (20, 3)]:
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(g0 % 4 == 0 and g1 % 4 == 0 for g0, g1 in r)


def test_two_rank_gather_matches_single_process(oracle):
    from mortal_amd import sharding

    n_games, world, port = 12, 2, 29517
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_games, q)) for r in range(world)]
    for p in procs:
        p.start()
    total, allsc = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _play(sharding.seeds_for((10000, KEY), 0, n_games))
    assert (allsc == ref).all()
    hist = [0, 0, 0, 0]
    for g in range(n_games):
        order = sorted(range(4), key=lambda s: -int(ref[g][s]))
        hist[order.index(g % 4)] += 1
    assert total == hist and sum(total) == n_games


# ---- the PRODUCT entry point under torch.distributed: libriichi.arena.OneVsThree.py_vs_py shards the games over the ranks
# and all-reduces the rank histogram (mortal_amd/arena.py); the device kernels run on the host emulator (tests/host/emu).
class _LowestLegalEngine:
    """The reference's engine contract (agent/mortal.rs:50-159) with a fixed policy: the lowest legal action id."""
    engine_type = "mortal"
    is_oracle = False
    version = 3
    enable_quick_eval = True
    enable_rule_based_agari_guard = False

    def __init__(self, name):
        self.name = name

    def react_batch(self, obs, masks, invisible_obs):
        import torch

        m = torch.as_tensor(np.stack(masks, axis=0))
        a = m.to(torch.uint8).argmax(dim=1)
        return a.tolist(), torch.zeros(m.shape, dtype=torch.float32).tolist(), m.tolist(), [True] * m.shape[0]


def _py_vs_py_worker(rank, world, port, seed_count, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "host")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import emu_pool
    from libriichi.arena import OneVsThree

    from mortal_amd import arena as A

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    A.BatchRunner.pool_cls = emu_pool.make_pool_class()
    got = OneVsThree(disable_progress_bar=True, deal_algo=0).py_vs_py(_LowestLegalEngine("a"), _LowestLegalEngine("b"),
                                                                      (10000, KEY), seed_count)
    q.put((rank, got))
    if world > 1:
        dist.destroy_process_group()


def test_sharded_py_vs_py_product_entry_point(oracle):
    """2 gloo ranks x OneVsThree.py_vs_py (8 hanchan): every rank returns the whole-run histogram, equal to the oracle's."""
    seed_count, world, port = 2, 2, 29531
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_py_vs_py_worker, args=(r, world, port, seed_count, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 4 * seed_count
    ref = _play([(10000 + g // 4, KEY) for g in range(n)])  # same policy on the oracle (deal_algo 0)
    hist = [0, 0, 0, 0]
    for g in range(n):
        order = sorted(range(4), key=lambda s: -int(ref[g][s]))
        hist[order.index(g % 4)] += 1
    assert got[0] == hist and got[1] == hist and sum(hist) == n
