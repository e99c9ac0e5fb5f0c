"""libriichi.mjai.Bot (mjai/bot.rs): one engine playing one seat over the mjai line protocol.

`Bot(engine, player_id).react(line, can_act=True) -> Optional[str]`: the event is applied to a device-backed PlayerState
(mortal_amd/state.py); when the seat can act, the obs/mask (and the kan-select row, agent/mortal.rs:244-287) go through
`engine.react_batch`, and the chosen action id becomes an mjai event with the device's action decoder (the same code
the arena's step kernel uses, agent/mortal.rs:338-573), including quick-eval, the rule-based agari guard and the
per-decision `meta` object (agent/mortal.rs:161-186,575-591).
"""
import json
import time

import numpy as np

from . import mjai_log
from .state import PlayerState, Q_AGARI_POINTS  # noqa: F401

Q_SCENE, Q_DECODE_ACTION = 6, 7


class Bot:
    def __init__(self, engine, player_id):
        if getattr(engine, "engine_type") != "mortal":
            raise NotImplementedError("Bot: only 'mortal' engines are supported")
        if not callable(getattr(engine, "react_batch", None)):
            raise TypeError("missing method react_batch")
        self.engine = engine
        self.name = str(engine.name)
        self.is_oracle = bool(engine.is_oracle)
        self.version = int(engine.version)
        self.enable_quick_eval = bool(engine.enable_quick_eval)
        self.enable_guard = bool(engine.enable_rule_based_agari_guard)
        if self.is_oracle:
            raise NotImplementedError("Bot cannot serve an oracle engine: a single seat has no invisible state")
        self.player_id = int(player_id)
        self.state = PlayerState(self.player_id)

    def _decode(self, action, kan_tile):
        out = self.state._query(Q_DECODE_ACTION, [int(action), int(kan_tile)])
        if out[2]:
            raise RuntimeError(f"failed to get reaction: action {action} is not legal here\nstate:\n{self.state.brief_info()}")
        w = (int(out[0]) & 0xFFFFFFFF) | ((int(out[1]) & 0xFFFFFFFF) << 32)
        if w == 0:
            return {"type": "none"}
        t = w & 15
        if t == mjai_log.LG_HORA:
            return {"type": "hora", "actor": (w >> 4) & 3, "target": (w >> 6) & 3}
        if t == mjai_log.LG_RYUKYOKU:
            return {"type": "ryukyoku"}
        return mjai_log.decode_events(np.array([w], dtype=np.uint64))[0]

    def _meta(self, q_row, mask_row, greedy):
        mask_bits = 0
        compact = []
        for i, (q, m) in enumerate(zip(q_row, mask_row)):
            if m:
                mask_bits |= 1 << i
                compact.append(float(q))
        return {"q_values": compact, "mask_bits": mask_bits, "is_greedy": bool(greedy), "shanten": self.state.shanten,
                "at_furiten": self.state.at_furiten}

    def react(self, line, /, *, can_act=True):
        try:
            data = json.loads(line) if isinstance(line, str) else dict(line)
        except json.JSONDecodeError as ex:
            raise ValueError(f"failed to parse event {line}") from ex
        ev_can_act = data.pop("can_act", None)
        data.pop("meta", None)
        cans = self.state.update(data)
        if not can_act or ev_can_act is False or not cans.can_act:
            return None
        t0 = time.perf_counter_ns()
        scene = self.state._query(Q_SCENE, [int(self.enable_quick_eval)])
        if scene[0] >= 0:  # quick eval: a single legal discard, no network call (agent/mortal.rs:210-242)
            pai = int(scene[0])
            return json.dumps({"type": "dahai", "actor": self.player_id, "pai": mjai_log.TILE_NAMES[pai],
                               "tsumogiri": int(scene[2]) == pai}, separators=(",", ":"))
        rows_obs, rows_mask = [], []
        kan_idx = None
        if scene[1]:
            o, m = self.state.encode_obs(self.version, True)
            kan_idx = len(rows_obs)
            rows_obs.append(o)
            rows_mask.append(m)
        o, m = self.state.encode_obs(self.version, False)
        main_idx = len(rows_obs)
        rows_obs.append(o)
        rows_mask.append(m)
        try:
            actions, q_values, masks_recv, is_greedy = self.engine.react_batch(rows_obs, rows_mask, None)
        except Exception as ex:
            raise RuntimeError(f"failed to execute `react_batch` on Python engine: {ex}") from ex
        action = int(actions[main_idx])
        if self.enable_guard and action == 43 and not self.state.rule_based_agari():
            q = np.array(q_values[main_idx], dtype=np.float32)
            q[43] = np.finfo(np.float32).min
            keys = q.view(np.int32).astype(np.int64)
            keys = np.where(keys < 0, keys ^ 0x7FFFFFFF, keys)  # f32::total_cmp order
            action = int(len(keys) - 1 - np.argmax(keys[::-1]))  # Iterator::max_by keeps the last maximum
        kan_tile = int(actions[kan_idx]) if kan_idx is not None else -1
        event = self._decode(action, kan_tile)
        meta = self._meta(q_values[main_idx], masks_recv[main_idx], is_greedy[main_idx])
        meta["batch_size"] = len(rows_obs)
        meta["eval_time_ns"] = time.perf_counter_ns() - t0
        if kan_idx is not None:
            meta["kan_select"] = self._meta(q_values[kan_idx], masks_recv[kan_idx], is_greedy[kan_idx])
        event["meta"] = meta
        return json.dumps(event, separators=(",", ":"))
