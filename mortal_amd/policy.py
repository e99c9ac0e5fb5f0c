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


class DeviceEngine:
    """Engine with the reference's duck-typed contract (agent/mortal.rs:53-74, mortal/engine.py:8-81) plus the
    device fast path `react_batch_device` that keeps actions on the GPU (no `.tolist()` round trip)."""

    engine_type = "mortal"

    def __init__(self, net, version, device, name="mortal_amd", enable_amp=True, enable_quick_eval=True,
                 max_batch=16384):
        self.net = net.to(device).eval()
        self.version = version
        self.device = torch.device(device)
        self.name = name
        self.is_oracle = False
        self.enable_amp = enable_amp
        self.enable_quick_eval = enable_quick_eval
        self.enable_rule_based_agari_guard = False
        self.max_batch = max_batch

    @torch.inference_mode()
    def react_batch_device(self, obs, masks):
        out = torch.empty(obs.shape[0], dtype=torch.int32, device=obs.device)
        for i in range(0, obs.shape[0], self.max_batch):  # bounded activation memory at 65k-row batches
            with torch.autocast("cuda", enabled=self.enable_amp):
                q = self.net(obs[i:i + self.max_batch], masks[i:i + self.max_batch])
            out[i:i + self.max_batch] = q.argmax(-1).to(torch.int32)
        return out

    def react_batch(self, obs, masks, invisible_obs):
        import numpy as np

        o = torch.as_tensor(np.stack(obs, axis=0), device=self.device)
        m = torch.as_tensor(np.stack(masks, axis=0), device=self.device)
        with torch.inference_mode(), torch.autocast(self.device.type, enabled=self.enable_amp):
            q = self.net(o, m)
        return q.argmax(-1).tolist(), q.tolist(), m.tolist(), [True] * o.shape[0]
