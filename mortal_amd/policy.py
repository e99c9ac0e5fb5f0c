"""Random-init policy/value network of the reference's architecture, for smoke tests and the `--policy brain` bench.

The production network is the reference's own `mortal/model.py` (Brain + DQN), which runs unchanged on the encoded
batch.  `/root/reference` does not exist on the GPU box, so benchmarks need a stand-in with the same shape and cost:
this module restates the v2..v4 architecture (model.py:10-231) — pre-activation 1-D ResNet over the 34 tile columns
with squeeze-style channel attention, BatchNorm + Mish, a 32-channel neck, a 1024-d feature and a dueling Q head over
the 46 actions.  Weights are random (there is no checkpoint to load offline).
"""
import torch
from torch import nn

from .pool import ACTION_SPACE, OBS_ROWS


class _ChannelGate(nn.Module):  # model.py:10-29
    def __init__(self, ch, ratio=16):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(ch, ch // ratio), nn.Mish(inplace=True), nn.Linear(ch // ratio, ch))

    def forward(self, x):
        w = (self.mlp(x.mean(-1)) + self.mlp(x.amax(-1))).sigmoid()
        return x * w.unsqueeze(-1)


class _PreActBlock(nn.Module):  # model.py:31-68 (pre_actv=True)
    def __init__(self, ch, eps):
        super().__init__()
        self.body = nn.Sequential(
            nn.BatchNorm1d(ch, momentum=0.01, eps=eps), nn.Mish(inplace=True), nn.Conv1d(ch, ch, 3, padding=1, bias=False),
            nn.BatchNorm1d(ch, momentum=0.01, eps=eps), nn.Mish(inplace=True), nn.Conv1d(ch, ch, 3, padding=1, bias=False),
        )
        self.gate = _ChannelGate(ch)

    def forward(self, x):
        return self.gate(self.body(x)) + x


class PolicyNet(nn.Module):
    """obs [B, C, 34] f32 + mask [B, 46] bool -> q [B, 46] (illegal actions = -inf)."""

    def __init__(self, version=4, conv_channels=192, num_blocks=40):
        super().__init__()
        assert version in (2, 3, 4)
        self.version = version
        eps = 1e-3 if version >= 3 else 1e-5
        ch = conv_channels
        self.trunk = nn.Sequential(
            nn.Conv1d(OBS_ROWS[version], ch, 3, padding=1, bias=False),
            *[_PreActBlock(ch, eps) for _ in range(num_blocks)],
            nn.BatchNorm1d(ch, momentum=0.01, eps=eps), nn.Mish(inplace=True),
            nn.Conv1d(ch, 32, 3, padding=1), nn.Mish(inplace=True), nn.Flatten(), nn.Linear(32 * 34, 1024),
            nn.Mish(inplace=True),
        )
        if version == 4:  # model.py:221-223
            self.head = nn.Linear(1024, 1 + ACTION_SPACE)
        else:
            hidden = 512 if version == 2 else 256
            self.v_head = nn.Sequential(nn.Linear(1024, hidden), nn.Mish(inplace=True), nn.Linear(hidden, 1))
            self.a_head = nn.Sequential(nn.Linear(1024, hidden), nn.Mish(inplace=True), nn.Linear(hidden, ACTION_SPACE))

    def forward(self, obs, mask):
        phi = self.trunk(obs)
        if self.version == 4:
            v, a = self.head(phi).split((1, ACTION_SPACE), dim=-1)
        else:
            v, a = self.v_head(phi), self.a_head(phi)
        a_mean = a.masked_fill(~mask, 0.0).sum(-1, keepdim=True) / mask.sum(-1, keepdim=True)
        return (v + a - a_mean).masked_fill(~mask, -torch.inf)  # model.py:225-231


def nucleus_sample(logits, top_p, generator=None):
    """One action per row from softmax(logits) restricted to its top-p nucleus (mortal/engine.py:83-94 `sample_top_p`):
    p >= 1 samples the whole distribution, p <= 0 is the argmax; otherwise the smallest set of most probable actions whose
    mass *before* the last member does not exceed p keeps its (unnormalised) probabilities, the rest get zero.
    -inf logits (illegal actions) have probability exactly 0 and are never drawn.  Stays on `logits.device`."""
    if top_p <= 0:
        return logits.argmax(-1)
    probs = logits.softmax(-1)
    if top_p < 1:
        order = probs.argsort(dim=-1, descending=True, stable=True)
        ranked = probs.gather(-1, order)
        mass_before = ranked.cumsum(-1) - ranked
        ranked = torch.where(mass_before > top_p, torch.zeros_like(ranked), ranked)
        pick = torch.multinomial(ranked, 1, generator=generator)
        return order.gather(-1, pick).squeeze(-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1)


def boltzmann_actions(q, masks, epsilon, temp, top_p, generator=None):
    """mortal/engine.py:72-81: with probability 1 - epsilon the greedy action, else a top-p sample of softmax(q / temp) over
    the legal actions.  Returns (actions int64 [B], is_greedy bool [B]) on q's device."""
    greedy = q.argmax(-1)
    if epsilon <= 0:
        return greedy, torch.ones(q.shape[0], dtype=torch.bool, device=q.device)
    is_greedy = torch.rand(q.shape[0], device=q.device, generator=generator) < (1 - epsilon)
    logits = (q.float() / temp).masked_fill(~masks, -torch.inf)
    return torch.where(is_greedy, greedy, nucleus_sample(logits, top_p, generator)), is_greedy


class DeviceEngine:
    """Engine with the reference's duck-typed contract (agent/mortal.rs:53-74, mortal/engine.py:8-81) plus the
    device fast path `react_batch_device` that keeps actions on the GPU (no `.tolist()` round trip): greedy, or the
    reference's Boltzmann-epsilon / top-p exploration (engine.py:72-94) sampled on the device."""

    engine_type = "mortal"

    def __init__(self, net, version, device, name="mortal_amd", enable_amp=True, enable_quick_eval=True,
                 max_batch=16384, boltzmann_epsilon=0, boltzmann_temp=1, top_p=1, enable_rule_based_agari_guard=False,
                 return_meta=False, seed=None, compile_net=False):
        self.net = net.to(device).eval()
        # compile_net: torch.compile (inductor) of the same module under the same autocast -- PyTorch's own graph compiler, measured
        # 7.1 x the eager forward on MI355X for the 192 x 40 net (tools/brain_tune.py).  Chunks then have ONE shape (max_batch rows,
        # the batch's ragged tail is padded in a staging buffer) so that nothing is compiled twice.
        self.compiled = bool(compile_net)
        self._stage = None
        if self.compiled:
            self.net = torch.compile(self.net)
        self.version = version
        self.device = torch.device(device)
        self.name = name
        self.is_oracle = False
        self.enable_amp = enable_amp
        self.enable_quick_eval = enable_quick_eval
        self.enable_rule_based_agari_guard = enable_rule_based_agari_guard
        self.max_batch = max_batch
        self.boltzmann_epsilon = boltzmann_epsilon
        self.boltzmann_temp = boltzmann_temp
        self.top_p = top_p
        self.return_meta = return_meta  # also return (q_values, is_greedy): the arena's log metadata / agari guard need them
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator(device=self.device)
            self.generator.manual_seed(seed)

    @torch.inference_mode()
    def react_batch_device(self, obs, masks):
        """-> actions int32 [B] on the device; with return_meta / the agari guard: (actions, q_values f32 [B, 46], is_greedy)."""
        n = obs.shape[0]
        out = torch.empty(n, dtype=torch.int32, device=obs.device)
        want_meta = self.return_meta or self.enable_rule_based_agari_guard
        q_all = torch.empty((n, ACTION_SPACE), dtype=torch.float32, device=obs.device) if want_meta else None
        greedy_all = torch.empty(n, dtype=torch.bool, device=obs.device) if want_meta else None
        for i in range(0, n, self.max_batch):  # bounded activation memory at 65k-row batches
            ob, mk = obs[i:i + self.max_batch], masks[i:i + self.max_batch]
            m = ob.shape[0]
            if self.compiled and m < self.max_batch:  # the ragged tail, padded to the one compiled shape (a legal action in every pad row)
                if self._stage is None or self._stage[0].shape[0] != self.max_batch or self._stage[0].shape[1:] != ob.shape[1:]:
                    self._stage = (torch.zeros((self.max_batch,) + tuple(ob.shape[1:]), dtype=ob.dtype, device=ob.device),
                                   torch.ones((self.max_batch, masks.shape[1]), dtype=torch.bool, device=ob.device))
                self._stage[0][:m] = ob
                self._stage[1][:m] = mk
                ob, mk = self._stage
            with torch.autocast(obs.device.type, enabled=self.enable_amp):
                q = self.net(ob, mk)
            q, mk = q[:m], mk[:m]
            act, is_greedy = boltzmann_actions(q, mk, self.boltzmann_epsilon, self.boltzmann_temp, self.top_p, self.generator)
            out[i:i + m] = act.to(torch.int32)
            if want_meta:
                q_all[i:i + m] = q.float()
                greedy_all[i:i + m] = is_greedy
        return (out, q_all, greedy_all) if want_meta else out

    def react_batch(self, obs, masks, invisible_obs):
        import numpy as np

        o = torch.as_tensor(np.stack(obs, axis=0), device=self.device)
        m = torch.as_tensor(np.stack(masks, axis=0), device=self.device)
        with torch.inference_mode(), torch.autocast(self.device.type, enabled=self.enable_amp):
            q = self.net(o, m)
        return q.argmax(-1).tolist(), q.tolist(), m.tolist(), [True] * o.shape[0]
