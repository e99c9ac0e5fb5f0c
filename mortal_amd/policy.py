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
                 return_meta=False, seed=None, compile_net="auto", min_bucket=256):
        self.net = net.to(device).eval()  # always the EAGER module: state_dict() / load_state_dict() keys stay the reference's
        # compile_net: the forward runs through torch.compile (PyTorch's inductor) of the same module under the same autocast --
        # measured 7.1 x the eager forward on MI355X for the 192 x 40 net (tools/brain_tune.py, bench.py workloads.brain_v4_compiled).
        #   "auto" (default): compiled on a GPU for production-sized batches (max_batch >= 1024), eager otherwise (CPU, small test
        #                     engines: compiling costs ~1 minute per distinct module); MORTAL_AMD_COMPILE_NET=0 / 1 overrides;
        #   True / False    : as said.
        # Chunks are padded to a few bucket sizes (powers of two from 256 rows up to max_batch) in a staging buffer, so a module is
        # compiled for at most log2(max_batch / 256) + 1 shapes and a 10-row call does not run a max_batch-row forward.  If the
        # compiler fails on the first call the engine says so once and continues EAGER (the net is PyTorch's either way).
        import os

        if compile_net == "auto":
            env = os.environ.get("MORTAL_AMD_COMPILE_NET")
            compile_net = (env == "1") if env in ("0", "1") else (torch.device(device).type == "cuda" and max_batch >= 1024)
        self.compiled = bool(compile_net)
        self._fwd = torch.compile(self.net, dynamic=False) if self.compiled else self.net
        self._stage = {}
        self.min_bucket = int(min_bucket)  # the smallest padded chunk of the compiled path (a power of two)
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
            b = self._bucket(m)
            if self.compiled and m < b:  # a ragged chunk, padded to its bucket's shape (a legal action in every pad row)
                st = self._stage.get(b)
                if st is None or st[0].shape[1:] != ob.shape[1:] or st[0].device != ob.device:
                    st = self._stage[b] = (torch.zeros((b,) + tuple(ob.shape[1:]), dtype=ob.dtype, device=ob.device),
                                           torch.ones((b, masks.shape[1]), dtype=torch.bool, device=ob.device))
                st[0][:m] = ob
                st[1][:m] = mk
                ob, mk = st
            q = self._forward(ob, mk)
            q, mk = q[:m], mk[:m]
            act, is_greedy = boltzmann_actions(q, mk, self.boltzmann_epsilon, self.boltzmann_temp, self.top_p, self.generator)
            out[i:i + m] = act.to(torch.int32)
            if want_meta:
                q_all[i:i + m] = q.float()
                greedy_all[i:i + m] = is_greedy
        return (out, q_all, greedy_all) if want_meta else out

    def warm(self, obs_rows=1012, buckets=None):
        """Compile every bucket shape now (compiled path only): with dynamic=False a chunk size seen for the first time costs ~10 s of
        inductor, and a batch's ragged tail changes bucket from one call to the next.  Call it once before a timed / latency-sensitive loop."""
        if not self.compiled:
            return
        if buckets is None:  # every bucket; or only the chunk sizes the caller knows its batches produce
            buckets, b = [], self.min_bucket
            while b <= self.max_batch:
                buckets.append(b)
                b *= 2
        for b in buckets:
            self.react_batch_device(torch.zeros((b, obs_rows, 34), dtype=torch.float32, device=self.device),
                                    torch.ones((b, 46), dtype=torch.bool, device=self.device))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _bucket(self, m):
        """The padded row count of an m-row chunk on the compiled path: the next power of two >= max(m, 256), at most max_batch."""
        b = self.min_bucket
        while b < m:
            b *= 2
        return min(b, self.max_batch)

    def _forward(self, ob, mk):
        try:
            with torch.autocast(ob.device.type, enabled=self.enable_amp):
                return self._fwd(ob, mk)
        except Exception as e:  # noqa: BLE001 - inductor / toolchain failures surface on the first call of a shape
            if not self.compiled:
                raise
            import warnings

            warnings.warn(f"DeviceEngine: torch.compile failed ({e!r:.300}); continuing with the eager module", RuntimeWarning)
            self.compiled = False
            self._fwd = self.net
            with torch.autocast(ob.device.type, enabled=self.enable_amp):
                return self.net(ob[:], mk[:])

    def react_batch(self, obs, masks, invisible_obs):
        import numpy as np

        o = torch.as_tensor(np.stack(obs, axis=0), device=self.device)
        m = torch.as_tensor(np.stack(masks, axis=0), device=self.device)
        with torch.inference_mode(), torch.autocast(self.device.type, enabled=self.enable_amp):
            q = self.net(o, m)
        return q.argmax(-1).tolist(), q.tolist(), m.tolist(), [True] * o.shape[0]
