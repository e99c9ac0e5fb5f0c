"""the device-side Boltzmann-epsilon / top-p sampling of `DeviceEngine.react_batch_device`
(mortal_amd/policy.py) against the reference's `MortalEngine._react_batch` exploration branch and `sample_top_p`
(mortal/engine.py:72-94): identical nucleus support, matching sampling frequencies, greedy flags, never an illegal action.
The reference functions are imported from /root/reference when present (here), else restated inline (the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch

from mortal_amd.policy import boltzmann_actions, nucleus_sample

REF = "/root/reference/mortal"


def _ref_sample_top_p():
    if os.path.exists(os.path.join(REF, "engine.py")):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "compat"), REF):
            if p not in sys.path:
                sys.path.append(p)
        import engine as ref_engine

        return ref_engine.sample_top_p

    def sample_top_p(logits, p):  # mortal/engine.py:83-94, restated for hosts without the reference tree
        from torch.distributions import Categorical

        if p >= 1:
            return Categorical(logits=logits).sample()
        if p <= 0:
            return logits.argmax(-1)
        probs = logits.softmax(-1)
        probs_sort, probs_idx = probs.sort(-1, descending=True)
        probs_sum = probs_sort.cumsum(-1)
        mask = probs_sum - probs_sort > p
        probs_sort[mask] = 0.0
        return probs_idx.gather(-1, probs_sort.multinomial(1)).squeeze(-1)

    return sample_top_p


def _fixture(rows=6, seed=3):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(rows, 46, generator=g) * 2.0
    masks = torch.rand(rows, 46, generator=g) < 0.35
    masks[:, 45] = True
    masks[0] = False
    masks[0, [3, 17]] = True  # two legal actions
    masks[1] = False
    masks[1, 9] = True        # a single legal action
    return q, masks


@pytest.mark.parametrize("top_p", [0.0, 0.3, 0.8, 1.0])
def test_nucleus_frequencies_match_reference(top_p):
    ref = _ref_sample_top_p()
    q, masks = _fixture()
    logits = (q / 0.7).masked_fill(~masks, -torch.inf)
    n = 40000
    big = logits.repeat_interleave(n, 0)
    torch.manual_seed(11)
    ours = nucleus_sample(big, top_p).view(-1, n)
    torch.manual_seed(12)
    theirs = ref(big.clone(), top_p).view(-1, n)
    for r in range(logits.shape[0]):
        fo = np.bincount(ours[r].numpy(), minlength=46) / n
        ft = np.bincount(theirs[r].numpy(), minlength=46) / n
        assert set(np.flatnonzero(fo)) <= set(np.flatnonzero(masks[r].numpy())), "illegal action sampled"
        # same nucleus: an action either side draws with visible frequency is drawn by the other one too
        assert set(np.flatnonzero(fo > 4e-3)) <= set(np.flatnonzero(ft > 1e-3)), (r, top_p)
        assert set(np.flatnonzero(ft > 4e-3)) <= set(np.flatnonzero(fo > 1e-3)), (r, top_p)
        assert np.abs(fo - ft).max() < 0.012, (r, top_p, np.abs(fo - ft).max())  # ~5 sigma at n = 40000
    if top_p <= 0:
        assert (ours == logits.argmax(-1, keepdim=True)).all()


def test_nucleus_support_is_the_reference_set():
    """Exact support: zero out like engine.py:89-92 and compare with the actions `nucleus_sample` can return."""
    q, masks = _fixture(rows=12, seed=5)
    logits = q.masked_fill(~masks, -torch.inf)
    for top_p in (0.2, 0.5, 0.9):
        probs = logits.softmax(-1)
        ps, idx = probs.sort(-1, descending=True)
        keep = ~((ps.cumsum(-1) - ps) > top_p) & (ps > 0)
        want = [set(idx[r][keep[r]].tolist()) for r in range(len(q))]
        torch.manual_seed(0)
        draws = nucleus_sample(logits.repeat_interleave(4000, 0), top_p).view(len(q), -1)
        for r in range(len(q)):
            got = set(draws[r].tolist())
            assert got <= want[r], (r, top_p)
            heavy = {a for a in want[r] if probs[r, a] > 0.01}
            assert heavy <= got, (r, top_p)


def test_boltzmann_epsilon_mixture():
    q, masks = _fixture(rows=4, seed=9)
    qm = q.masked_fill(~masks, -torch.inf)
    a, g = boltzmann_actions(qm, masks, 0, 1, 1)
    assert g.all() and (a == qm.argmax(-1)).all()
    n = 50000
    torch.manual_seed(1)
    a, g = boltzmann_actions(qm.repeat_interleave(n, 0), masks.repeat_interleave(n, 0), 0.25, 0.5, 0.9)
    a, g = a.view(4, n), g.view(4, n)
    assert abs(float(g.float().mean()) - 0.75) < 0.01                      # bernoulli(1 - epsilon), engine.py:73
    assert (a[g] == qm.argmax(-1, keepdim=True).expand(4, n)[g]).all()     # greedy rows take the argmax
    assert masks.gather(1, a).all()                                        # never an illegal action
    # the explored rows follow softmax(q / temp) restricted to the nucleus
    logits = (qm / 0.5)
    for r in range(4):
        probs = logits[r].softmax(-1)
        ps, idx = probs.sort(descending=True)
        keep = ~((ps.cumsum(-1) - ps) > 0.9) & (ps > 0)
        want = torch.zeros(46)
        want[idx[keep]] = ps[keep] / ps[keep].sum()
        got = np.bincount(a[r][~g[r]].numpy(), minlength=46) / max(1, int((~g[r]).sum()))
        assert np.abs(got - want.numpy()).max() < 0.02


def test_device_engine_returns_meta_for_logs():
    """react_batch_device with return_meta: (actions, q_values, is_greedy) on the batch's device."""
    from mortal_amd.policy import DeviceEngine, PolicyNet

    torch.manual_seed(0)
    eng = DeviceEngine(PolicyNet(version=3, conv_channels=16, num_blocks=1), 3, "cpu", enable_amp=False, boltzmann_epsilon=0.5,
                       boltzmann_temp=1.0, top_p=0.9, return_meta=True, seed=7)
    obs = torch.rand(32, 934, 34)
    masks = torch.rand(32, 46) < 0.3
    masks[:, 45] = True
    act, qv, greedy = eng.react_batch_device(obs, masks)
    assert act.dtype == torch.int32 and qv.shape == (32, 46) and greedy.dtype == torch.bool
    assert masks.gather(1, act.long().unsqueeze(1)).all()
    assert (act[greedy].long() == qv.argmax(-1)[greedy]).all() and not greedy.all() and greedy.any()


def test_device_engine_compiled_path_pads_the_ragged_tail(monkeypatch):
    """DeviceEngine(compile_net=True) (bench.py workloads.brain_v4_compiled): every chunk the compiled module sees has ONE shape
    (max_batch rows; the batch's ragged tail is padded in a staging buffer, pad rows with every action legal) and the actions of
    the real rows equal the eager engine's.  torch.compile itself is PyTorch's; here it is replaced by a shape-recording wrapper."""
    import torch

    from mortal_amd.policy import DeviceEngine, PolicyNet

    seen = []

    class Recorder(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, obs, mask):
            seen.append(tuple(obs.shape))
            return self.m(obs, mask)

    monkeypatch.setattr(torch, "compile", lambda m, **kw: Recorder(m))
    torch.manual_seed(3)
    net = PolicyNet(version=4, conv_channels=16, num_blocks=2)
    eager = DeviceEngine(net, 4, "cpu", enable_amp=False, max_batch=64)
    comp = DeviceEngine(net, 4, "cpu", enable_amp=False, max_batch=64, compile_net=True)
    obs = torch.rand(150, 1012, 34)
    mask = torch.rand(150, 46) < 0.3
    mask[:, 45] = True
    a_e = eager.react_batch_device(obs, mask)
    a_c, q_c, g_c = DeviceEngine(net, 4, "cpu", enable_amp=False, max_batch=64, compile_net=True, return_meta=True).react_batch_device(obs, mask)
    assert torch.equal(a_e, comp.react_batch_device(obs, mask)) and torch.equal(a_e, a_c)
    assert q_c.shape == (150, 46) and g_c.all()
    assert seen and all(s == (64, 1012, 34) for s in seen), seen  # 64 + 64 + (22 padded to 64)
    assert mask[torch.arange(150), a_e.long()].all()  # legal actions only
    # padding goes to a few bucket sizes, not to max_batch: a 10-row call of a 16,384-row engine runs a 256-row forward
    seen.clear()
    big = DeviceEngine(net, 4, "cpu", enable_amp=False, max_batch=1024, compile_net=True)
    for rows in (10, 150, 256, 700, 1024):
        assert torch.equal(big.react_batch_device(obs[:1].expand(rows, -1, -1).contiguous(), mask[:1].expand(rows, -1).contiguous()),
                           a_e[:1].expand(rows))
    assert [s[0] for s in seen] == [256, 256, 256, 1024, 1024], seen
    assert set(big._stage) == {256, 1024}
    # the engine's module stays the eager one: checkpoints saved / loaded through engine.net keep the reference's parameter names
    assert not any(k.startswith("_orig_mod.") for k in big.net.state_dict()) and big.net is net
    # "auto" never compiles on the CPU; the environment switch forces either way
    assert DeviceEngine(net, 4, "cpu", enable_amp=False).compiled is False
    monkeypatch.setenv("MORTAL_AMD_COMPILE_NET", "1")
    assert DeviceEngine(net, 4, "cpu", enable_amp=False).compiled is True
    monkeypatch.setenv("MORTAL_AMD_COMPILE_NET", "0")
    assert DeviceEngine(net, 4, "cpu", enable_amp=False, max_batch=4096).compiled is False


def test_device_engine_compile_failure_falls_back_to_eager_loudly(monkeypatch):
    """A compiler that throws on the first call (no inductor toolchain, an unsupported op) costs a RuntimeWarning, not the run."""
    import torch

    from mortal_amd.policy import DeviceEngine, PolicyNet

    class Broken(torch.nn.Module):
        def forward(self, obs, mask):
            raise RuntimeError("inductor: no C++ compiler")

    monkeypatch.setattr(torch, "compile", lambda m, **kw: Broken())
    torch.manual_seed(4)
    net = PolicyNet(version=4, conv_channels=16, num_blocks=1)
    obs = torch.rand(20, 1012, 34)
    mask = torch.rand(20, 46) < 0.3
    mask[:, 45] = True
    want = DeviceEngine(net, 4, "cpu", enable_amp=False, compile_net=False).react_batch_device(obs, mask)
    eng = DeviceEngine(net, 4, "cpu", enable_amp=False, compile_net=True)
    with pytest.warns(RuntimeWarning, match="torch.compile failed"):
        got = eng.react_batch_device(obs, mask)
    assert torch.equal(got, want) and eng.compiled is False
