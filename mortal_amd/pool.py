"""TablePool: the SoA pool of concurrent tables in HBM, driven through the C-ABI (include/mortal_amd.h).

PyTorch is used only for device memory and the HIP stream; all compute is in libmortal_amd.so.
Reference counterpart: `BatchGame::run` (libriichi/src/arena/game.rs:230-316).
"""
import ctypes as C

import numpy as np
import torch

from . import tables
from ._lib import MortalAmdError, check, lib

OBS_ROWS = {1: 938, 2: 942, 3: 934, 4: 1012}
ACTION_SPACE = 46
DEAL_RAND08, DEAL_RAND09 = 0, 1  # include/mortal_amd.h MJ_DEAL_*


def default_deal_algo():
    """The wall shuffle of `Board::init_from_seed` (arena/board.rs:99-109) depends on the `rand` crate generation.
    Default = rand 0.9.1, the version the reference's Cargo.lock pins (Cargo.lock:1042-1043), so the same (seed, key)
    gives today's reference's games; MORTAL_AMD_DEAL_ALGO=rand08 selects the older Fisher-Yates (the shuffle of the
    reference's published example log)."""
    import os

    v = os.environ.get("MORTAL_AMD_DEAL_ALGO", "rand09").strip().lower()
    if v in ("0", "rand08", "rand0.8", "0.8"):
        return DEAL_RAND08
    if v in ("1", "rand09", "rand0.9", "0.9", "0.9.1"):
        return DEAL_RAND09
    raise ValueError(f"MORTAL_AMD_DEAL_ALGO={v!r}: expected rand08 or rand09")

_tables_ready = False


def _ensure_tables():
    global _tables_ready
    if not _tables_ready:
        p = tables.payload()
        check(lib.mj_tables_upload(p, len(p)))
        _tables_ready = True


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class TablePool:
    _L = lib  # the C-ABI library (tests/host/emu_pool.py substitutes the host emulation of the same sources)

    def _stream(self):
        return _stream()

    def _bind_device(self, device):
        if not torch.cuda.is_available():
            raise MortalAmdError("TablePool needs a HIP device (torch.cuda.is_available() is False)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        _ensure_tables()

    def __init__(self, n_tables, version=4, deal_algo=None, device="cuda:0", max_rows=0):
        self._bind_device(device)
        self.n_tables = n_tables
        self.version = version
        self.C = OBS_ROWS[version]
        self.max_rows = max_rows or 8 * n_tables
        self.deal_algo = default_deal_algo() if deal_algo is None else int(deal_algo)
        self.h = self._L.mj_pool_create(n_tables, version, self.deal_algo, self.max_rows)
        if not self.h:
            raise MortalAmdError(self._L.mj_last_error().decode())
        self.n_rows = [0, 0]
        self.n_games_total = n_tables
        self.versions = [version, version]

    def close(self):
        if getattr(self, "h", None):
            self._L.mj_pool_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, seeds, game_ids=None, agent_of_seat=None, n_games_total=None):
        """seeds: list of (nonce, key) per table; agent_of_seat: uint8 per table, bit s = agent of seat s."""
        assert len(seeds) == self.n_tables
        nonces = np.ascontiguousarray([s[0] for s in seeds], dtype=np.uint64)
        keys = np.ascontiguousarray([s[1] for s in seeds], dtype=np.uint64)
        gid = np.ascontiguousarray(game_ids if game_ids is not None else np.arange(self.n_tables), dtype=np.uint32)
        aos = np.ascontiguousarray(agent_of_seat if agent_of_seat is not None else np.zeros(self.n_tables),
                                   dtype=np.uint8)
        self.n_games_total = int(n_games_total or (int(gid.max()) + 1))
        check(self._L.mj_pool_reset(self.h, nonces.ctypes.data, keys.ctypes.data, gid.ctypes.data, aos.ctypes.data,
                                self.n_games_total))
        self.n_rows = [0, 0]

    def configure(self, agent, enable_quick_eval=True, enable_rule_based_agari_guard=False, version=0):
        check(self._L.mj_pool_configure(self.h, agent, int(version), int(enable_quick_eval),
                                    int(enable_rule_based_agari_guard)))
        if version:
            self.versions[agent] = version

    def enable_log(self, words_per_table=16384):
        """Turn the per-table mjai event log on (call before reset/step).  16384 words cover ~60 kyoku."""
        check(self._L.mj_pool_enable_log(self.h, int(words_per_table)))
        self.log_cap = int(words_per_table)

    def read_logs(self, chunk=1024):
        """-> list (one per table) of uint64 arrays holding the event words logged so far."""
        n = self.n_tables
        lens = np.zeros(n, dtype=np.uint32)
        check(self._L.mj_log_lengths(self.h, lens.ctypes.data, self._stream()))
        if (lens > self.log_cap).any():
            raise MortalAmdError(f"event log overflow on table {int(np.argmax(lens > self.log_cap))}")
        out = []
        for t0 in range(0, n, chunk):
            k = min(chunk, n - t0)
            buf = np.empty((k, self.log_cap), dtype=np.uint64)
            check(self._L.mj_log_read(self.h, t0, k, buf.ctypes.data, self._stream()))
            out += [buf[i, :lens[t0 + i]].copy() for i in range(k)]
        return out

    # ---- log replay (dataset loader)
    def replay_load(self, scripts, tracked, always_include_kan_select=True, nonces=None, keys=None):
        """scripts: one uint64 word array per table (mjai_log.encode_events); tracked: 4-bit seat mask per table;
        nonces / keys: per-table game seeds for scripts that ask the device to rebuild the wall (trust_seed)."""
        assert len(scripts) == self.n_tables
        off = np.zeros(len(scripts) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(x) for x in scripts])
        script = np.ascontiguousarray(np.concatenate(scripts) if len(scripts) else np.zeros(0), dtype=np.uint64)
        tr = np.ascontiguousarray(tracked, dtype=np.uint8)
        n64 = np.ascontiguousarray(nonces, dtype=np.uint64) if nonces is not None else None
        k64 = np.ascontiguousarray(keys, dtype=np.uint64) if keys is not None else None
        check(self._L.mj_replay_load(self.h, script.ctypes.data, off.ctypes.data, tr.ctypes.data, len(scripts),
                                 int(always_include_kan_select), n64.ctypes.data if n64 is not None else None,
                                 k64.ctypes.data if k64 is not None else None))

    def replay_step(self):
        check(self._L.mj_replay_step(self.h, self._stream()))
        out = (C.c_int32 * 2)()
        check(self._L.mj_rows_count(self.h, out, self._stream()))
        self.n_rows = [out[0], out[1]]
        return out[0]

    def replay_meta(self):
        """int32 cuda [n, 8] = label, log, seat, kyoku index, turn, shanten, is-kan-row, event index."""
        n = self.n_rows[0]
        meta = torch.empty((n, 8), dtype=torch.int32, device=self.device)
        if n:
            check(self._L.mj_replay_meta(self.h, meta.data_ptr(), self._stream()))
        return meta

    def set_refill(self, nonce_stride):
        check(self._L.mj_pool_set_refill(self.h, nonce_stride))

    def set_start_stagger(self, cycles):
        """Steady-state runs: table t starts its first hanchan at cycle hash(t) % cycles (after reset + set_refill)."""
        check(self._L.mj_pool_set_start_stagger(self.h, int(cycles), self._stream()))

    def step(self, actions0=None, actions1=None, q0=None, q1=None, ev0=None, ev1=None):
        """One arena cycle; actionsN = int32 cuda tensor with one action per row of agent N's last batch; qN = that
        batch's q-values (f32 cuda [n, 46]), needed only by an agent configured with the rule-based agari guard."""
        for a in (actions0, actions1):
            if a is not None:
                assert a.device.type == self.device.type and a.dtype == torch.int32 and a.is_contiguous()
        for q in (q0, q1):
            if q is not None:
                assert q.device.type == self.device.type and q.dtype == torch.float32 and q.is_contiguous() and q.shape[-1] == 46
        # one action per row of the batch the previous step produced: the step kernel indexes these tensors by row id
        for ag, (a, q) in enumerate(((actions0, q0), (actions1, q1))):
            if a is not None and a.numel() != self.n_rows[ag]:
                raise MortalAmdError(f"agent {ag}: {a.numel()} actions for a batch of {self.n_rows[ag]} rows")
            if q is not None and q.numel() != self.n_rows[ag] * ACTION_SPACE:
                raise MortalAmdError(f"agent {ag}: q-values of shape {tuple(q.shape)} for a batch of {self.n_rows[ag]} rows")
            ev = (ev0, ev1)[ag]
            if ev is not None and (ev.dtype != torch.int64 or ev.numel() != self.n_rows[ag] or not ev.is_contiguous()
                                   or ev.device.type != self.device.type):
                raise MortalAmdError(f"agent {ag}: reactions must be {self.n_rows[ag]} contiguous int64 event words on the device")
            if a is None and ev is None and self.n_rows[ag]:
                raise MortalAmdError(f"agent {ag}: the previous batch had {self.n_rows[ag]} rows but no actions were passed")
        p0 = actions0.data_ptr() if actions0 is not None and actions0.numel() else None
        p1 = actions1.data_ptr() if actions1 is not None and actions1.numel() else None
        pq0 = q0.data_ptr() if q0 is not None and q0.numel() else None
        pq1 = q1.data_ptr() if q1 is not None and q1.numel() else None
        pe0 = ev0.data_ptr() if ev0 is not None and ev0.numel() else None
        pe1 = ev1.data_ptr() if ev1 is not None and ev1.numel() else None
        check(self._L.mj_step_ev(self.h, p0, p1, pq0, pq1, pe0, pe1, self._stream()))
        out = (C.c_int32 * 2)()
        check(self._L.mj_rows_count(self.h, out, self._stream()))
        self.n_rows = [out[0], out[1]]
        return self.n_rows

    def rows(self, agent):
        """Row descriptors of the current batch as an int64 cpu array [n, 3] = (table, seat, is_kan)."""
        n = self.n_rows[agent]
        ptr = self._L.mj_rows_dev(self.h, agent)
        d = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            self._copy_rows(d.data_ptr(), ptr, 4 * n)
        v = d.cpu().numpy().view(np.uint32)
        return np.stack([v & 0x0FFFFFFF, (v >> 28) & 3, v >> 31], axis=1).astype(np.int64)

    def _copy_rows(self, dst, src, nbytes):
        torch.cuda.current_stream().synchronize()
        _memcpy_d2d(dst, src, nbytes)

    def encode(self, agent, obs=None, masks=None):
        """Encode agent's rows in place into (or into fresh) device tensors. Returns (obs [n,C,34] f32, masks [n,46] bool)."""
        n = self.n_rows[agent]
        if obs is None:
            obs = torch.empty((n, OBS_ROWS[self.versions[agent]], 34), dtype=torch.float32, device=self.device)
        if masks is None:
            masks = torch.empty((n, ACTION_SPACE), dtype=torch.bool, device=self.device)
        assert obs.is_contiguous() and masks.is_contiguous() and obs.shape[0] >= n and masks.shape[0] >= n
        if n:
            check(self._L.mj_encode(self.h, agent, obs.data_ptr(), masks.data_ptr(), self._stream()))
        return obs[:n], masks[:n]

    def encode_oracle(self, agent, out=None):
        """Invisible ("oracle") obs of agent's rows (board.rs:679-782): f32 cuda [n, 211|217, 34]."""
        n = self.n_rows[agent]
        R = self._L.mj_oracle_obs_rows(self.versions[agent])
        if out is None:
            out = torch.empty((n, R, 34), dtype=torch.float32, device=self.device)
        assert out.device.type == self.device.type and out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= n
        assert tuple(out.shape[1:]) == (R, 34)
        check(self._L.mj_encode_oracle(self.h, agent, out.data_ptr(), self._stream()))
        return out[:n]

    def random_policy(self, agent, masks, seed, cycle, out=None):
        n = self.n_rows[agent]
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            check(self._L.mj_random_policy(self.h, agent, masks.data_ptr(), seed, cycle, out.data_ptr(), self._stream()))
        return out[:n]

    def greedy_policy(self, agent, masks, obs, seed, cycle, out=None):
        """Tenpai-seeking policy on device (mj_greedy_policy); obs = the encoded batch of this agent's rows."""
        n = self.n_rows[agent]
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            check(self._L.mj_greedy_policy(self.h, agent, masks.data_ptr(), obs.data_ptr(), seed, cycle, out.data_ptr(),
                                           self._stream()))
        return out[:n]

    def counters(self):
        out = (C.c_uint64 * 8)()
        check(self._L.mj_counters(self.h, out, self._stream()))
        return dict(steps=out[0], games=out[1], errors=out[2], decisions=out[3], quick=out[4], cycles=out[5],
                    sp_overflow=out[6])

    def results(self):
        scores = np.zeros((self.n_games_total, 4), dtype=np.int32)
        done = np.zeros(self.n_games_total, dtype=np.uint8)
        check(self._L.mj_results(self.h, scores.ctypes.data, done.ctypes.data, self._stream()))
        return scores, done

    def first_error(self):
        t = C.c_int(-1)
        code = check(self._L.mj_pool_first_error(self.h, C.byref(t), self._stream()))
        return code, t.value

    def encode_timing(self, enable=True):
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(self._L.mj_encode_timing(self.h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sp_timing(self):
        """(total ms, launches) of the SP-table kernel since the last call (recorded while encode timing is on)."""
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(self._L.mj_sp_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sp_phase_ticks(self):
        """Cumulative mj_k_sp phase timers (workgroup ticks, 100 MHz) and counts since the pool was created."""
        out = (C.c_uint64 * 8)()
        check(self._L.mj_sp_phase_ticks(self.h, out, self._stream()))
        return dict(zip(("overflow", "rows", "setup", "expand", "level0", "eval", "write", "states"), (int(x) for x in out)))

    def set_sp_schedule(self, mode=-2, max_rows=0, wide_grid=0, min_level1=0, min_level2=0):
        """Small-pool schedule of the SP kernel (include/mortal_amd.h mj_pool_set_sp_schedule): mode -1 auto / 0 never / 1 always."""
        check(self._L.mj_pool_set_sp_schedule(self.h, mode, max_rows, wide_grid, min_level1, min_level2))

    def sp_schedule_stats(self):
        out = (C.c_uint64 * 4)()
        check(self._L.mj_sp_schedule_stats(self.h, out, self._stream()))
        return dict(zip(("hybrid_launches", "rows_promoted", "rows_swept", "wide_gave_up"), (int(x) for x in out)))

    def debug_table(self, table):
        size = self._L.mj_debug_table_size()
        buf = (C.c_uint8 * size)()
        check(self._L.mj_debug_table(self.h, table, buf, size, self._stream()))
        raw = bytes(buf)
        out = {}
        for ent in self._L.mj_debug_layout().decode().strip(";").split(";"):
            name, size, count, off = ent.split(":")
            size, count, off = int(size), int(count), int(off)
            dt = {1: np.uint8, 2: np.uint16, 4: np.int32, 8: np.uint64}[size]
            out[name] = np.frombuffer(raw, dtype=dt, count=count, offset=off).copy()
        return out


def _memcpy_d2d(dst, src, nbytes):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(dst, src, nbytes, 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise MortalAmdError(f"hipMemcpy failed: {rc}")
