"""Host side of the batched arena: mirrors `libriichi.arena` (reference libriichi/src/arena/one_vs_three.rs:17-113,
two_vs_two.rs:17-110) on top of the HIP table pool.

The poll/commit cycle of `BatchGame::run` (arena/game.rs:286-304) becomes, per cycle:
    mj_step (commit + poll + classify, on device)  ->  per agent: mj_encode -> engine.react_batch -> actions
The engine contract is the reference's (agent/mortal.rs:50-159): attributes `engine_type`, `name`, `is_oracle`,
`version`, `enable_quick_eval`, `enable_rule_based_agari_guard`, and
`react_batch(obs, masks, invisible_obs) -> (actions, q_values, masks, is_greedy)`.

Zero-copy: `MortalEngine._react_batch` (mortal/engine.py:53-55) does `torch.as_tensor(np.stack(obs, axis=0), device=..)`.
We pass `[proxy]` whose `__array_function__` answers `np.stack` with the pre-stacked device tensor, so the unchanged
engine consumes the encoded batch in place.  Engines that define `react_batch_device(obs, masks)` (our extension,
returns a device int tensor of actions) skip the `.tolist()` round trip as well.
"""
import os
import time

import numpy as np
import torch

from .pool import ACTION_SPACE, OBS_ROWS, TablePool
from ._lib import MortalAmdError, check


class _StackedBatch:
    """Stands in for a list of per-row numpy arrays; np.stack([proxy], axis=0) returns the device tensor."""

    def __init__(self, tensor):
        self.tensor = tensor

    def __array_function__(self, func, types, args, kwargs):
        if func is np.stack:
            return self.tensor
        return NotImplemented

    def __len__(self):
        return self.tensor.shape[0]


def _check_engine(engine):
    """agent/py_agent.rs:24-37: dispatch on `engine_type`."""
    et = getattr(engine, "engine_type")
    if et == "mjai-log":  # agent/mjai_log.rs:34-63
        for method in ("react_batch", "start_game", "end_kyoku", "end_game"):
            if not callable(getattr(engine, method, None)):
                raise TypeError(f"missing method {method}")
        # every decision gets a row (no quick-eval), nothing is encoded for it, reactions are explicit mjai events
        return dict(name=str(engine.name), version=0, quick=False, guard=False, oracle=False, mjai_log=True)
    if et != "mortal":
        raise ValueError(f"unknown engine type {et}")
    if not callable(getattr(engine, "react_batch", None)):
        raise TypeError("missing method react_batch")
    return dict(name=str(engine.name), version=int(engine.version), quick=bool(engine.enable_quick_eval),
                guard=bool(getattr(engine, "enable_rule_based_agari_guard")), oracle=bool(getattr(engine, "is_oracle")),
                mjai_log=False)


class GameState:
    """What an mjai-log engine's `react_batch` receives per decision (agent/mjai_log.rs:22-32)."""

    def __init__(self, game_index, state, events_json):
        self.game_index = game_index    # index of this player in the list given to `set_player_ids`
        self.state = state              # libriichi.state.PlayerState (a copy of the seat's state)
        self.events_json = events_json  # the current kyoku's log so far, a JSON array of mjai events


def pack_reaction(ev):
    """mjai reaction dict -> the header word of the device's event format (mj_state.h LG_*), 0 for {"type":"none"}."""
    from . import mjai_log as ML

    t = ev["type"]
    if t == "none":
        return 0
    if t in ("hora", "ryukyoku"):  # header only: deltas / ura markers are the board's business
        code = ML.LG_HORA if t == "hora" else ML.LG_RYUKYOKU
        return code | (int(ev.get("actor", 0)) << 4) | (int(ev.get("target", 0)) << 6)
    return int(ML.encode_events([ev])[0])


class BatchRunner:
    """Drives N tables to completion with up to two engines (agent 0 / agent 1)."""

    pool_cls = TablePool  # the test suite substitutes the host emulation of the same kernels (tests/host/emu_pool.py)

    def __init__(self, engines, seeds, agent_of_seat, device=None, deal_algo=None, keep_log=False):
        self.engines = engines
        self.cfg = [_check_engine(e) for e in engines]
        dev = device
        if dev is None:
            d0 = getattr(engines[0], "device", None)
            dev = d0 if isinstance(d0, torch.device) and d0.type == "cuda" else torch.device(
                f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0")
        self.device = torch.device(dev)
        n = len(seeds)
        versions = [c["version"] for c in self.cfg if c["version"]] or [3]
        for c in self.cfg:
            c["version"] = c["version"] or versions[0]  # an mjai-log agent's rows are never encoded
        self.pool = self.pool_cls(n, version=self.cfg[0]["version"], deal_algo=deal_algo, device=str(self.device))
        self.device = self.pool.device
        self.any_mjai_log = any(c["mjai_log"] for c in self.cfg)
        if keep_log or self.any_mjai_log:  # mjai-log engines read the kyoku's events from the device log
            self.pool.enable_log()
        self.seeds = list(seeds)
        self.agent_of_seat = np.asarray(agent_of_seat, dtype=np.uint8)
        self.pool.reset(seeds, game_ids=np.arange(n), agent_of_seat=agent_of_seat, n_games_total=n)
        for a, c in enumerate(self.cfg):
            self.pool.configure(a, enable_quick_eval=c["quick"], version=c["version"], enable_rule_based_agari_guard=c["guard"])
        if len(engines) == 1:
            self.pool.configure(1, enable_quick_eval=self.cfg[0]["quick"], version=self.cfg[0]["version"],
                                enable_rule_based_agari_guard=self.cfg[0]["guard"])
        # player planning of the reference (one_vs_three.rs:140-191, two_vs_two.rs:138-190): per agent the seats it plays,
        # games in order, seats ascending; an mjai-log engine addresses its players by their index in that list
        self.player_index = [{}, {}]
        player_ids = [[], []]
        for g in range(n):
            for seat in range(4):
                a = (int(self.agent_of_seat[g]) >> seat) & 1 if len(engines) > 1 else 0
                self.player_index[a][(g, seat)] = len(player_ids[a])
                player_ids[a].append(seat)
        self._kyoku_ends = {}  # game -> end_kyoku events already reported to the mjai-log engines
        self._log_cache = {}   # game -> (words decoded, events): the mjai-log agents' view of the device event log
        self._lens, self._lens_cycle = None, -1
        for a, eng in enumerate(engines):
            if self.cfg[a]["mjai_log"]:
                if callable(getattr(eng, "set_player_ids", None)):
                    eng.set_player_ids(player_ids[a])
                for idx in range(len(player_ids[a])):
                    eng.start_game(idx)
        self.cycles = 0
        self.keep_log = keep_log
        self.meta_batches = {}  # step index -> per agent (q_values, masks, is_greedy, eval_time_ns) of the rows it commits

    def _policy(self, agent, obs, masks, invisible=None):
        """-> (actions int32 cuda [n], q_values f32 cuda [n,46] or None).  q-values are kept only for a guarded agent."""
        eng = self.engines[agent]
        guard = self.cfg[agent]["guard"]
        if hasattr(eng, "react_batch_device"):
            out = eng.react_batch_device(obs, masks, invisible) if invisible is not None else eng.react_batch_device(obs, masks)
            out = out if isinstance(out, tuple) else (out,)
            act, q, is_greedy = out[0], (out[1] if len(out) > 1 else None), (out[2] if len(out) > 2 else None)
            if guard and q is None:
                raise RuntimeError("enable_rule_based_agari_guard: react_batch_device must return (actions, q_values)")
            if self.keep_log and q is not None:  # per-decision metadata of the game logs (agent/mortal.rs:161-186)
                g = is_greedy if is_greedy is not None else torch.ones(obs.shape[0], dtype=torch.bool)
                self._last_meta = (q.float().cpu().numpy().reshape(-1, ACTION_SPACE), masks.cpu().numpy().astype(bool),
                                   g.cpu().numpy().astype(bool).reshape(-1))
            q = q.to(device=self.device, dtype=torch.float32).contiguous() if guard else None
            return act.to(device=self.device, dtype=torch.int32).contiguous(), q
        try:
            inv = [_StackedBatch(invisible)] if invisible is not None else None  # mortal.rs:137-145
            actions, q_values, _m, _g = eng.react_batch([_StackedBatch(obs)], [_StackedBatch(masks)], inv)
        except Exception as ex:  # same context string as agent/mortal.rs:149
            raise RuntimeError(f"failed to execute `react_batch` on Python engine: {ex}") from ex
        if len(actions) != obs.shape[0]:
            raise RuntimeError("react_batch returned a batch of the wrong size")
        if self.keep_log:  # per-decision metadata of the game logs (agent/mortal.rs:161-186)
            self._last_meta = (np.asarray(q_values, dtype=np.float32).reshape(-1, ACTION_SPACE),
                               np.asarray(_m, dtype=bool).reshape(-1, ACTION_SPACE), np.asarray(_g, dtype=bool).reshape(-1))
        q = torch.as_tensor(q_values, dtype=torch.float32, device=self.device).contiguous() if guard else None
        return torch.as_tensor(actions, dtype=torch.int32, device=self.device), q

    def _fail(self, code, tbl):
        """Illegal action / rule violation on table `tbl`: the reference aborts the whole run with the offending event and
        the seat's `brief_info()` (arena/board.rs:524-533); here the table's state at the time of the check is dumped."""
        lines = [f"table {tbl} (seed {self.seeds[tbl] if 0 <= tbl < len(self.seeds) else '?'}): illegal action or rule violation "
                 f"(error code {code}, include/mortal_amd.h MJ_ERR_*)"]
        try:
            d = self.pool.debug_table(tbl)
            names = "1m 2m 3m 4m 5m 6m 7m 8m 9m 1p 2p 3p 4p 5p 6p 7p 8p 9p 1s 2s 3s 4s 5s 6s 7s 8s 9s E S W N P F C".split()
            lines.append(f"  kyoku {int(d['kyoku'][0])} honba {int(d['honba'][0])} kyotaku {int(d['kyotaku'][0])} scores "
                         f"{d['scores'].tolist()} tiles_left {int(d['tiles_left'][0])} pending seats {int(d['pending'][0]):04b}")
            for seat in range(4):
                mp, sz = int(d["hand_mp"][seat]), int(d["hand_sz"][seat])
                cnt = [(mp >> (3 * t)) & 7 for t in range(18)] + [(sz >> (3 * t)) & 7 for t in range(16)]
                hand = " ".join(names[t] for t in range(34) for _ in range(cnt[t]))
                lines.append(f"  seat {seat}: shanten {int(d['shanten'][seat])} cans 0x{int(d['cans'][seat]):x} "
                             f"last_self_tsumo {int(d['last_self_tsumo'][seat])} hand [{hand}]")
        except Exception as ex:  # noqa: BLE001 - the diagnostic must not hide the error itself
            lines.append(f"  (state dump unavailable: {ex})")
        return MortalAmdError("\n".join(lines))

    def _log_lengths(self):
        """Event-log lengths of every table: read ONCE per cycle (the mjai-log agents of a cycle share it)."""
        if self._lens_cycle != self.cycles:
            lens = np.zeros(self.pool.n_tables, dtype=np.uint32)
            check(self.pool._L.mj_log_lengths(self.pool.h, lens.ctypes.data, self.pool._stream()))
            self._lens, self._lens_cycle = lens, self.cycles
        return self._lens

    def _game_log(self, g):
        """The decoded event log of game g, kept across cycles: only the words appended since the last look are decoded (a log
        always ends on an event boundary), instead of the whole hanchan every cycle."""
        from . import mjai_log

        n = int(self._log_lengths()[g])
        if n > self.pool.log_cap:
            raise MortalAmdError(f"event log overflow on table {g}")
        done, events = self._log_cache.get(g, (0, []))
        if n > done:
            buf = np.empty((1, self.pool.log_cap), dtype=np.uint64)
            check(self.pool._L.mj_log_read(self.pool.h, int(g), 1, buf.ctypes.data, self.pool._stream()))
            events.extend(mjai_log.decode_events(buf[0, done:n]))  # in place: callers only read the list
            self._log_cache[g] = (n, events)
        return events

    def _report_kyoku_ends(self, g, events):
        """end_kyoku(player index) for every kyoku of game g that has ended since the last look (game.rs:113-118)."""
        n_end = sum(1 for e in events if e["type"] == "end_kyoku")
        for _ in range(n_end - self._kyoku_ends.get(g, 0)):
            for a, eng in enumerate(self.engines):
                if self.cfg[a]["mjai_log"]:
                    for seat in range(4):
                        idx = self.player_index[a].get((g, seat))
                        if idx is not None:
                            eng.end_kyoku(idx)
        self._kyoku_ends[g] = n_end

    def _mjai_log_policy(self, agent):
        """agent/mjai_log.rs:65-150: GameState per acting seat -> engine.react_batch -> validated, packed reactions."""
        import json

        from .state import PlayerState

        eng = self.engines[agent]
        rows = self.pool.rows(agent)
        words = np.zeros(len(rows), dtype=np.int64)
        states, where = [], []
        logs = {}
        for r, (g, seat, is_kan) in enumerate(rows):
            if is_kan:
                continue  # kan-select rows belong to the mortal agent's protocol; the event names the tile itself
            g, seat = int(g), int(seat)
            if g not in logs:
                logs[g] = self._game_log(g)
                self._report_kyoku_ends(g, logs[g])
            evs = logs[g]
            start = max(i for i, e in enumerate(evs) if e["type"] == "start_kyoku")
            st = PlayerState.view(self.pool, g, seat)
            states.append(GameState(self.player_index[agent][(g, seat)], st, json.dumps(evs[start:], separators=(",", ":"))))
            where.append((r, st))
        try:
            raw = eng.react_batch(states) if states else []
        except Exception as ex:
            raise RuntimeError(f"failed to execute `react_batch` on Python engine: {ex}") from ex
        if len(raw) != len(states):
            raise RuntimeError("react_batch returned a batch of the wrong size")
        for (r, st), text in zip(where, raw):
            ev = json.loads(text)
            try:  # BoardState::step validates every reaction (board.rs:524-533)
                st.validate_reaction(ev)
            except ValueError as ex:
                raise MortalAmdError(f"invalid action: {ev}: {ex}\nstate:\n{st.brief_info()}") from ex
            words[r] = pack_reaction(ev)
        return torch.from_numpy(words).to(self.device)

    def run(self, max_cycles=1 << 30, progress=None):
        """progress: None = silent (disable_progress_bar), else a label: the reference's bar message
        (`cycles: N (x cycle/s), actions: M (y action/s)`, arena/game.rs:303-311) goes to stderr about once a second."""
        import sys

        pool = self.pool
        acts = [None, None]
        qs = [None, None]
        evs = [None, None]
        n_games = pool.n_tables
        t_start = t_last = time.perf_counter()

        def report(final=False):
            c = pool.counters()
            secs = max(time.perf_counter() - t_start, 1e-9)
            print(f"\r{progress + ' ' if progress else ''}{c['games']}/{n_games} games  cycles: {self.cycles} "
                  f"({self.cycles / secs:.3f} cycle/s), actions: {c['steps']} ({c['steps'] / secs:.3f} action/s)",
                  end="\n" if final else "", file=sys.stderr, flush=True)

        while True:
            if self.cycles >= max_cycles:
                raise MortalAmdError("max_cycles exceeded")
            n = pool.step(acts[0], acts[1], qs[0], qs[1], evs[0], evs[1])
            self.cycles += 1
            code, tbl = pool.first_error() if (self.cycles & 63) == 0 else (0, -1)
            if code:
                raise self._fail(code, tbl)
            if progress is not None and time.perf_counter() - t_last > 1.0:
                t_last = time.perf_counter()
                report()
            acts = [None, None]
            qs = [None, None]
            evs = [None, None]
            if n[0] == 0 and n[1] == 0:
                c = pool.counters()
                if c["games"] >= n_games:
                    break
                continue
            for a in (0, 1):
                if n[a] == 0:
                    continue
                if self.cfg[min(a, len(self.cfg) - 1)]["mjai_log"]:
                    evs[a] = self._mjai_log_policy(min(a, len(self.cfg) - 1))
                    continue
                obs, masks = pool.encode(a)
                inv = pool.encode_oracle(a) if self.cfg[min(a, len(self.cfg) - 1)]["oracle"] else None
                self._last_meta = None
                t0 = time.perf_counter_ns()
                acts[a], qs[a] = self._policy(a, obs, masks, inv)
                if acts[a].numel() != n[a] or (qs[a] is not None and tuple(qs[a].shape) != (n[a], ACTION_SPACE)):
                    raise RuntimeError(f"engine returned {acts[a].numel()} actions"
                                       + (f" / q-values of shape {tuple(qs[a].shape)}" if qs[a] is not None else "")
                                       + f" for a batch of {n[a]} rows")
                if self.keep_log and self._last_meta is not None:
                    # these rows are committed by the NEXT mj_step call, whose index the device writes into the log tags
                    self.meta_batches.setdefault(self.cycles, {})[a] = self._last_meta + (time.perf_counter_ns() - t0,)
        if progress is not None:
            report(final=True)
        code, tbl = pool.first_error()
        if code:
            raise self._fail(code, tbl)
        if pool.counters()["sp_overflow"]:
            raise MortalAmdError("obs v4: a decision's single-player state graph exceeded the device scratch capacity "
                                 "(SP_CAP in mortal_amd/csrc/mj_sp.hip); its SP planes would be incomplete")
        scores, done = pool.results()
        if not (done == 1).all():
            raise MortalAmdError("some games did not finish")
        if self.any_mjai_log:  # the last kyoku's end_kyoku, then end_game(index, scores) (game.rs:113-118,199-201)
            for g in range(n_games):
                self._report_kyoku_ends(g, self._game_log(g))
                for a, eng in enumerate(self.engines):
                    if self.cfg[a]["mjai_log"]:
                        for seat in range(4):
                            idx = self.player_index[a].get((g, seat))
                            if idx is not None:
                                eng.end_game(idx, [int(x) for x in scores[g]])
        return scores

    @staticmethod
    def _meta(batch, row, tag, with_batch=False):
        """Metadata of one decision (agent/mortal.rs:161-186 gen_meta + :575-591), keys in the order of mjai::Metadata."""
        q, m, greedy, eval_ns = batch
        bits = 0
        for i in np.flatnonzero(m[row]):
            bits |= 1 << int(i)
        meta = {"q_values": [float(x) for x in q[row][m[row]]], "mask_bits": bits, "is_greedy": bool(greedy[row])}
        if with_batch:
            meta["batch_size"] = int(q.shape[0])
            meta["eval_time_ns"] = int(eval_ns)
        meta["shanten"] = tag["shanten"]
        meta["at_furiten"] = tag["at_furiten"]
        return meta

    def dump_logs(self, log_dir, splits):
        """Write one mjai `.json.gz` per game (result.rs:32-51, one_vs_three.rs:195-225); `splits` = files per seed."""
        from . import mjai_log

        os.makedirs(log_dir, exist_ok=True)
        names_of_agent = [c["name"] for c in self.cfg]
        if len(names_of_agent) == 1:
            names_of_agent = names_of_agent * 2
        paths = []
        for g, words in enumerate(self.pool.read_logs()):
            names = [names_of_agent[(int(self.agent_of_seat[g]) >> s) & 1] for s in range(4)]  # game.rs:186
            tags = []
            events = mjai_log.decode_events(words, tags)
            for ev, tag in zip(events, tags):
                if tag is None:
                    continue
                agent = (int(self.agent_of_seat[g]) >> ev.get("actor", 0)) & 1
                batch = self.meta_batches.get(tag["cycle"], {}).get(agent)
                if batch is None:
                    continue  # device-side engine (react_batch_device): no q-values came back to the host
                meta = self._meta(batch, tag["row"], tag, with_batch=True)
                if tag["kan_row"] is not None:
                    meta["kan_select"] = self._meta(batch, tag["kan_row"], tag)
                ev["meta"] = meta
            name = f"{self.seeds[g][0]}_{self.seeds[g][1]}_{'abcd'[g % splits]}.json.gz"
            paths.append(mjai_log.write_game_log_as(os.path.join(log_dir, name), names, self.seeds[g], events))
        return paths

    def close(self):
        self.pool.close()


def _rank_by_player(scores):
    """rankings.rs:8-21: stable sort by -score; ties favour the lower seat."""
    order = sorted(range(4), key=lambda i: -int(scores[i]))
    rank = [0] * 4
    for r, pid in enumerate(order):
        rank[pid] = r
    return rank


class OneVsThree:
    """libriichi.arena.OneVsThree (arena/one_vs_three.rs:17-113)."""

    def __init__(self, *, disable_progress_bar=False, log_dir=None, deal_algo=None):
        self.disable_progress_bar = disable_progress_bar
        self.log_dir = log_dir
        self.deal_algo = deal_algo  # None = pool.default_deal_algo() (rand 0.9.1 unless MORTAL_AMD_DEAL_ALGO says otherwise)

    def py_vs_py(self, challenger, champion, seed_start, seed_count):
        """Returns the rank histogram [1st, 2nd, 3rd, 4th] of the challenger over seed_count*4 hanchan."""
        from . import sharding

        n = int(seed_count) * 4
        # One process per GPU (torch.distributed initialised, e.g. under torchrun): rank r plays the contiguous game range
        # [g0, g1) — multiples of 4, so the four seat rotations of a seed stay together — with its own pool and its own
        # copy of the engines; the only collective is the all-reduce of the 4-entry rank histogram,
        # so every rank returns the reference's whole-run result (one_vs_three.rs:55-60 sums it in-process).
        rank, world, backend = sharding.dist_info()
        g0, g1 = sharding.shard_range(n, rank, world) if world > 1 else (0, n)
        seeds = [(int(seed_start[0]) + g // 4, int(seed_start[1])) for g in range(g0, g1)]  # one_vs_three.rs:140-142
        # challenger (agent 0) sits at seat g % 4, the champion (agent 1) on the other three (one_vs_three.rs:144-191)
        aos = np.array([0xF & ~(1 << (g % 4)) for g in range(g0, g1)], dtype=np.uint8)
        rankings = [0, 0, 0, 0]
        dev = None
        if g1 > g0:
            runner = BatchRunner([challenger, champion], seeds, aos, keep_log=self.log_dir is not None, deal_algo=self.deal_algo)
            dev = runner.device
            try:
                scores = runner.run(progress=None if self.disable_progress_bar else f"rank {rank}" if world > 1 else "")
                if self.log_dir is not None:
                    runner.dump_logs(self.log_dir, 4)
            finally:
                runner.close()
            for i, g in enumerate(range(g0, g1)):
                rankings[_rank_by_player(scores[i])[g % 4]] += 1  # one_vs_three.rs:55-60
        if world > 1:
            red_dev = dev if (backend == "nccl" and dev is not None) else torch.device(
                f"cuda:{torch.cuda.current_device()}") if backend == "nccl" else torch.device("cpu")
            rankings = sharding.allreduce_rank_histogram(rankings, device=red_dev)
        return rankings

    def ako_vs_py(self, engine, seed_start, seed_count):
        raise NotImplementedError("akochan agents are out of scope (SURVEY.md §2 row 2)")

    def py_vs_ako(self, engine, seed_start, seed_count):
        raise NotImplementedError("akochan agents are out of scope (SURVEY.md §2 row 2)")


class TwoVsTwo:
    """libriichi.arena.TwoVsTwo (arena/two_vs_two.rs:17-110): seed_count*2 hanchan, returns None."""

    def __init__(self, *, disable_progress_bar=False, log_dir=None, deal_algo=None):
        self.disable_progress_bar = disable_progress_bar
        self.log_dir = log_dir
        self.deal_algo = deal_algo  # None = pool.default_deal_algo() (rand 0.9.1 unless MORTAL_AMD_DEAL_ALGO says otherwise)

    def py_vs_py(self, challenger, champion, seed_start, seed_count):
        from . import sharding

        n = int(seed_count) * 2
        rank, world, _backend = sharding.dist_info()  # one process per GPU: contiguous seed ranges, nothing to reduce
        g0, g1 = sharding.shard_range(n, rank, world, group=2) if world > 1 else (0, n)
        seeds = [(int(seed_start[0]) + g // 2, int(seed_start[1])) for g in range(g0, g1)]  # two_vs_two.rs:138-140
        # split A: challenger at seats 0,2; split B: 1,3 (two_vs_two.rs:142-172)
        aos = np.array([0b1010 if g % 2 == 0 else 0b0101 for g in range(g0, g1)], dtype=np.uint8)
        if g1 > g0:
            runner = BatchRunner([challenger, champion], seeds, aos, keep_log=self.log_dir is not None, deal_algo=self.deal_algo)
            try:
                self.last_scores = runner.run(progress=None if self.disable_progress_bar else "")
                if self.log_dir is not None:
                    runner.dump_logs(self.log_dir, 2)
            finally:
                runner.close()
        return None

    def ako_vs_py(self, *a, **k):
        raise NotImplementedError("akochan agents are out of scope")

    py_vs_ako = py_vs_ako_one = ako_vs_py


__all__ = ["OneVsThree", "TwoVsTwo", "BatchRunner", "ACTION_SPACE", "OBS_ROWS"]
