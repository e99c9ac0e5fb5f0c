"""-m gpu: the libriichi-compatible arena (OneVsThree.py_vs_py) end to end with a random-init policy net, both through
the legacy `react_batch` list contract (zero-copy proxy) and the device fast path, cross-checked against the oracle
arena driven by the same network (BASELINE configs[0]-style plumbing, 8 hanchan)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEY = 0xD5DFAA4CEF265CD7


def _engine(version, seed, name, device_path):
    import torch

    from mortal_amd.policy import DeviceEngine, PolicyNet

    torch.manual_seed(seed)
    net = PolicyNet(version=version, conv_channels=32, num_blocks=2)
    eng = DeviceEngine(net, version, "cuda:0", name=name, enable_amp=False, compile_net=False)  # (bit-equal q-values against the oracle loop)
    if not device_path:
        eng.react_batch_device = None
        del eng.react_batch_device
        # instance attribute deletion does not hide the class method: wrap instead
        class Legacy:  # only the reference's duck-typed surface
            pass
        leg = Legacy()
        for k in ("engine_type", "name", "is_oracle", "version", "enable_quick_eval", "enable_rule_based_agari_guard",
                  "device"):
            setattr(leg, k, getattr(eng, k))
        leg.react_batch = eng.react_batch
        return leg, eng
    return eng, eng


def _oracle_rankings(oracle, challenger, champion, seed_start, seed_count, version):
    """The reference's BatchGame loop on the oracle, with the same two networks deciding."""
    import torch

    n = seed_count * 4
    seeds = [(seed_start[0] + g // 4, seed_start[1]) for g in range(n)]
    from mortal_amd.pool import default_deal_algo

    arena = oracle.Arena(seeds, deal_algo=default_deal_algo(), enable_quick_eval=True, version=version, keep_log=False)
    while arena.n_live > 0:
        rows = arena.poll()
        k = len(rows)
        obs, masks = arena.encode(0, k, want_obs=True)
        act = np.full(k, 45, dtype=np.int32)
        if k:
            is_chal = (rows[:, 1] == rows[:, 0] % 4)
            for eng, sel in ((challenger, is_chal), (champion, ~is_chal)):
                if sel.any():
                    o = torch.from_numpy(obs[sel]).cuda()
                    m = torch.from_numpy(masks[sel].astype(bool)).cuda()
                    act[sel] = eng.react_batch_device(o, m).cpu().numpy()
        arena.commit(act)
    scores = np.array([arena.result(g)[0] for g in range(n)])
    rankings = [0, 0, 0, 0]
    for g in range(n):
        order = sorted(range(4), key=lambda i: -int(scores[g][i]))
        rankings[order.index(g % 4)] += 1
    return rankings, scores


@pytest.mark.parametrize("device_path", [False, True])
def test_one_vs_three_py_vs_py_matches_oracle(oracle, device_path):
    from libriichi.arena import OneVsThree

    version = 3
    chal, chal_dev = _engine(version, 1, "challenger", device_path)
    cham, cham_dev = _engine(version, 2, "champion", device_path)
    env = OneVsThree(disable_progress_bar=True)
    got = env.py_vs_py(challenger=chal, champion=cham, seed_start=(10000, KEY), seed_count=2)
    assert sum(got) == 8
    want, _ = _oracle_rankings(oracle, chal_dev, cham_dev, (10000, KEY), 2, version)
    assert got == want


def test_two_vs_two_matches_oracle_loop(oracle):
    """TwoVsTwo.py_vs_py returns None like the reference (two_vs_two.rs:17-110); the hanchan it played are compared with the
    oracle's BatchGame loop under the reference's plan (two_vs_two.rs:138-190): two games per seed, split A = challenger at seats
    0 and 2, split B = seats 1 and 3, the same two networks deciding."""
    import torch

    from libriichi.arena import TwoVsTwo
    from mortal_amd.pool import default_deal_algo

    a, _ = _engine(3, 3, "a", True)
    b, _ = _engine(3, 4, "b", True)
    env = TwoVsTwo(disable_progress_bar=True)
    seed_start, seed_count = (20000, KEY), 3
    assert env.py_vs_py(a, b, seed_start, seed_count) is None
    n = seed_count * 2
    seeds = [(seed_start[0] + g // 2, seed_start[1]) for g in range(n)]
    arena = oracle.Arena(seeds, deal_algo=default_deal_algo(), enable_quick_eval=True, version=3, keep_log=False)
    while arena.n_live > 0:
        rows = arena.poll()
        k = len(rows)
        obs, masks = arena.encode(0, k, want_obs=True)
        act = np.full(k, 45, dtype=np.int32)
        if k:
            is_chal = (rows[:, 1] % 2) == (rows[:, 0] % 2)  # split A (even game): seats 0, 2; split B: seats 1, 3
            for eng, sel in ((a, is_chal), (b, ~is_chal)):
                if sel.any():
                    o = torch.from_numpy(obs[sel]).cuda()
                    m = torch.from_numpy(masks[sel].astype(bool)).cuda()
                    act[sel] = eng.react_batch_device(o, m).cpu().numpy()
        arena.commit(act)
    want = np.array([arena.result(g)[0] for g in range(n)])
    assert np.array_equal(np.asarray(env.last_scores), want), (env.last_scores, want)


def test_is_oracle_and_guard_engine_contract(oracle):
    """An engine with is_oracle=True receives `invisible_obs` as a list whose np.stack is the [B, 217, 34] device batch
    (agent/mortal.rs:137-145); one with enable_rule_based_agari_guard=True has its q-values routed to the step kernel."""
    import torch

    from libriichi.arena import OneVsThree

    seen = {}

    class Eng:
        engine_type = "mortal"
        name = "oracle-guard"
        is_oracle = True
        version = 3
        enable_quick_eval = True
        enable_rule_based_agari_guard = True

        def react_batch(self, obs, masks, invisible_obs):
            o = np.stack(obs, axis=0)
            m = np.stack(masks, axis=0)
            inv = np.stack(invisible_obs, axis=0)
            assert isinstance(inv, torch.Tensor) and inv.is_cuda and inv.shape == (o.shape[0], 217, 34)
            seen["rows"] = seen.get("rows", 0) + int(o.shape[0])
            # first legal action; q-values favour it, -inf on illegal ones (engine.py masks the same way)
            q = torch.where(m, torch.linspace(1.0, 0.0, 46, device=m.device).expand_as(m), torch.tensor(-torch.inf, device=m.device))
            a = q.argmax(dim=1)
            return a.tolist(), q.tolist(), m.tolist(), [True] * o.shape[0]

    e = Eng()
    got = OneVsThree(disable_progress_bar=True).py_vs_py(challenger=e, champion=e, seed_start=(10000, KEY), seed_count=1)
    assert sum(got) == 4 and seen["rows"] > 0


def test_log_dir_and_stat(oracle, tmp_path):
    """player.py's flow (mortal/player.py:60-73): OneVsThree(log_dir=...).py_vs_py(...) then Stat.from_dir(dir, name):
    the rank histogram of the challenger from the logs equals py_vs_py's return value."""
    from libriichi.arena import OneVsThree
    from libriichi.stat import Stat

    chal, _ = _engine(3, 1, "challenger", True)
    cham, _ = _engine(3, 2, "champion", True)
    d = str(tmp_path / "logs")
    got = OneVsThree(disable_progress_bar=True, log_dir=d).py_vs_py(challenger=chal, champion=cham, seed_start=(10000, KEY), seed_count=2)
    import os

    files = sorted(os.listdir(d))
    assert files == sorted(f"{10000 + i}_{KEY}_{s}.json.gz" for i in range(2) for s in "abcd")
    st = Stat.from_dir(d, "challenger", True)
    assert st.game == 8 and [st.rank_1, st.rank_2, st.rank_3, st.rank_4] == got
    ch = Stat.from_dir(d, "champion", True)
    assert ch.game == 24 and st.point + ch.point == 0


def test_log_meta_objects(oracle, tmp_path):
    """With the legacy react_batch engines the dumped logs carry the reference's per-decision `meta` objects
    (agent/mortal.rs:161-186,575-591): compact q-values over the mask, mask bits, greedy flag, batch size, shanten and
    furiten of the acting seat — and the events themselves are unchanged."""
    import gzip
    import json
    import os

    from libriichi.arena import OneVsThree

    chal, _ = _engine(3, 1, "challenger", False)
    cham, _ = _engine(3, 2, "champion", False)
    d = str(tmp_path / "logs")
    OneVsThree(disable_progress_bar=True, log_dir=d).py_vs_py(challenger=chal, champion=cham, seed_start=(10000, KEY), seed_count=1)
    n_meta = n_kan = 0
    for f in sorted(os.listdir(d)):
        evs = [json.loads(l) for l in gzip.open(os.path.join(d, f), "rt")]
        ps = [oracle.PlayerState(p) for p in range(4)]
        at_decision = [None] * 4  # (mask, shanten, furiten) of each seat when it last could act
        for ev in evs[1:-1]:
            meta = ev.pop("meta", None)
            if meta is not None:
                a = ev["actor"]
                mask, shanten, furiten = at_decision[a]
                assert list(meta)[:3] == ["q_values", "mask_bits", "is_greedy"] and meta["batch_size"] >= 1
                assert meta["shanten"] == shanten and meta["at_furiten"] == furiten, (ev, meta)
                bits = sum(1 << int(i) for i in np.flatnonzero(mask))
                assert meta["mask_bits"] == bits and len(meta["q_values"]) == int(mask.sum())
                n_meta += 1
                n_kan += "kan_select" in meta
            for a, p in enumerate(ps):
                cans = p.update(ev)
                # a reach_accepted logged between a discard and the call answering it does not renew the decision
                if ev["type"] != "reach_accepted" and any(v for k, v in cans.items() if k != "target_actor"):
                    sn = p.snapshot()
                    at_decision[a] = (p.encode_obs(3, False)[1], sn["shanten"], sn["at_furiten"])
    assert n_meta > 500


def test_mjai_log_engine_on_device(oracle):
    """engine_type 'mjai-log' on the real device: explicit mjai reactions through mj_step_ev, GameState / callbacks — the same
    scenario tests/test_mjai_log_agent.py runs on the emulator (ExampleMjaiLogEngine vs a lowest-legal 'mortal' engine)."""
    import test_mjai_log_agent as T
    from libriichi.arena import OneVsThree, TwoVsTwo

    from mortal_amd.pool import default_deal_algo

    eng = T._example_engine_cls()("logger")
    got = OneVsThree(disable_progress_bar=True).py_vs_py(challenger=eng, champion=T._Lowest(), seed_start=(10000, KEY), seed_count=2)
    sc = T._oracle_scores(oracle, [(10000 + g // 4, KEY) for g in range(8)], lambda g, s: s == g % 4, default_deal_algo())
    want = [0, 0, 0, 0]
    for g in range(8):
        order = sorted(range(4), key=lambda i: -int(sc[g][i]))
        want[order.index(g % 4)] += 1
    assert got == want and eng.player_ids == [0, 1, 2, 3] * 2
    env = TwoVsTwo(disable_progress_bar=True)
    env.py_vs_py(challenger=T._example_engine_cls()("logger"), champion=T._Lowest(), seed_start=(20000, KEY), seed_count=2)
    sc = T._oracle_scores(oracle, [(20000 + g // 2, KEY) for g in range(4)], lambda g, s: (s % 2 == 0) == (g % 2 == 0), default_deal_algo())
    assert (env.last_scores == sc).all()


def test_device_engine_exploration_meta_on_device(oracle, tmp_path):
    """react_batch_device with Boltzmann-epsilon / top-p sampling on the GPU, q-values and is_greedy flowing into the log meta."""
    import gzip
    import json
    import os

    import torch
    from libriichi.arena import OneVsThree

    from mortal_amd.policy import DeviceEngine, PolicyNet

    def engine(seed, name, eps):
        torch.manual_seed(seed)
        return DeviceEngine(PolicyNet(version=4, conv_channels=32, num_blocks=2), 4, "cuda:0", name=name, enable_amp=False,
                            boltzmann_epsilon=eps, top_p=0.9, return_meta=True, seed=seed, compile_net=False)

    d = str(tmp_path / "logs")
    got = OneVsThree(disable_progress_bar=True, log_dir=d).py_vs_py(challenger=engine(1, "challenger", 0.25), champion=engine(2, "champion", 0.0),
                                                                    seed_start=(10000, KEY), seed_count=2)
    assert sum(got) == 8
    n_meta = n_explored = 0
    for f in sorted(os.listdir(d)):
        for line in gzip.open(os.path.join(d, f), "rt"):
            meta = json.loads(line).get("meta")
            if meta:
                n_meta += 1
                n_explored += not meta["is_greedy"]
                assert len(meta["q_values"]) == bin(meta["mask_bits"]).count("1")
    assert n_meta > 1000 and 0 < n_explored < n_meta / 2


def test_device_engine_real_torch_compile_matches_eager():
    """DeviceEngine's DEFAULT on a GPU (compile_net="auto": torch.compile of the module, dynamic=False, autocast inside the call,
    bucketed padding of ragged chunks) against the eager module on real encoded-shape batches: q-values within fp32 reassociation
    noise, the same greedy action wherever the best two q-values are not a near-tie, a legal action everywhere; the engine's module
    keeps the reference's parameter names (the only test of the compiled path had replaced torch.compile)."""
    import torch

    from mortal_amd.policy import DeviceEngine, PolicyNet

    torch.manual_seed(11)
    net = PolicyNet(version=4, conv_channels=16, num_blocks=1)
    comp = DeviceEngine(net, 4, "cuda:0", enable_amp=False, max_batch=1024, return_meta=True)
    assert comp.compiled, "auto = compiled on a GPU for max_batch >= 1024"
    eager = DeviceEngine(net, 4, "cuda:0", enable_amp=False, max_batch=1024, return_meta=True, compile_net=False)
    g = torch.Generator(device="cuda").manual_seed(5)
    for rows in (1500, 37):  # 1024 + 476 (padded to 512); 37 (padded to 256)
        obs = torch.rand((rows, 1012, 34), device="cuda", generator=g)
        mask = torch.rand((rows, 46), device="cuda", generator=g) < 0.3
        mask[:, 45] = True
        a_c, q_c, _ = comp.react_batch_device(obs, mask)
        a_e, q_e, _ = eager.react_batch_device(obs, mask)
        assert comp.compiled, "torch.compile fell back to eager on this box"
        fin = torch.isfinite(q_e)
        assert torch.equal(fin, torch.isfinite(q_c)) and torch.equal(fin, mask)
        assert torch.allclose(q_c[fin], q_e[fin], rtol=1e-4, atol=1e-4)
        top2 = q_e.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert clear.float().mean() > 0.5 and torch.equal(a_c[clear], a_e[clear])
        assert mask.gather(1, a_c.long().unsqueeze(1)).all()
    assert not any(k.startswith("_orig_mod.") for k in comp.net.state_dict())
