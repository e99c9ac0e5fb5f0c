"""ctypes binding of oracle/libmjoracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libmjoracle.so")

TILE_NAMES = (
    [f"{n}{s}" for s in "mps" for n in range(1, 10)] + ["E", "S", "W", "N", "P", "F", "C", "5mr", "5pr", "5sr", "?"]
)
TILE_ID = {s: i for i, s in enumerate(TILE_NAMES)}

EV_TYPES = [
    "none", "start_game", "start_kyoku", "tsumo", "dahai", "chi", "pon", "daiminkan", "kakan", "ankan", "dora",
    "reach", "reach_accepted", "hora", "ryukyoku", "end_kyoku", "end_game",
]
EV_ID = {s: i for i, s in enumerate(EV_TYPES)}
EV_INTS = 82

CANS = ["can_discard", "can_chi_low", "can_chi_mid", "can_chi_high", "can_pon", "can_daiminkan", "can_kakan",
        "can_ankan", "can_riichi", "can_tsumo_agari", "can_ron_agari", "can_ryukyoku", "target_actor"]

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in os.listdir(ORACLE_DIR) if f.endswith((".cc", ".h", "Makefile"))
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    # bench.py's cpu_baseline workers load the -O3 -march=native build of the same sources (oracle/Makefile: native), built on the
    # measurement host; everything else uses the portable library
    L = C.CDLL(os.environ.get("MJ_ORACLE_LIB") or LIB_PATH)
    L.mjo_last_error.restype = C.c_char_p
    L.mjo_ps_new.restype = C.c_void_p
    L.mjo_ps_clone.restype = C.c_void_p
    L.mjo_ps_clone.argtypes = [C.c_void_p]
    L.mjo_ps_free.argtypes = [C.c_void_p]
    L.mjo_arena_new.restype = C.c_void_p
    L.mjo_arena_new.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mjo_arena_player_state.restype = C.c_void_p
    L.mjo_arena_steps.restype = C.c_long
    L.mjo_arena_cycles.restype = C.c_long
    for name in ["mjo_arena_free", "mjo_arena_poll", "mjo_arena_n_live", "mjo_arena_steps", "mjo_arena_cycles"]:
        getattr(L, name).argtypes = [C.c_void_p]
    L.mjo_arena_rows.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_arena_restart.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    L.mjo_arena_park.argtypes = [C.c_void_p]
    L.mjo_arena_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mjo_arena_commit.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_arena_encode_oracle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mjo_arena_commit_q.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjo_arena_guard_hits.restype = C.c_long
    L.mjo_arena_guard_hits.argtypes = [C.c_void_p]
    L.mjo_arena_result.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.mjo_arena_done_flags.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_arena_game_view.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.mjo_arena_player_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mjo_arena_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.mjo_ps_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjo_ps_validate_reaction.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_ps_encode_obs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mjo_ps_set_tehai.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.mjo_ps_call.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mjo_ps_agari_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.mjo_ps_snapshot.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_ps_uncond_tenpai.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_ps_kawa.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.mjo_ps_set_scores.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_ps_get_rank.argtypes = [C.c_void_p, C.c_void_p]
    L.mjo_ps_scene.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.mjo_ps_decode_action.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mjo_ps_sp_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mjo_sp_calc.argtypes = [C.c_void_p] * 7 + [C.c_int]
    L.mjo_set_tables.argtypes = [C.c_void_p, C.c_size_t]
    L.mjo_calc_shanten.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mjo_agari.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    L.mjo_check_ankan_after_riichi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.mjo_point.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mjo_sha3_256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.mjo_chacha12.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.mjo_deal.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    from mortal_amd import tables

    p = tables.payload()
    if L.mjo_set_tables(p, len(p)) != 0:
        raise RuntimeError(L.mjo_last_error().decode())
    _lib = L
    return L


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise OracleError(lib().mjo_last_error().decode())
    return rc


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- hands (hand.rs:14-71, tenhou.net/2 format)
def hand_with_aka(s):
    ret = np.zeros(37, dtype=np.uint8)
    stack = []
    for ch in s:
        if ch.isdigit():
            stack.append(int(ch))
        elif ch in "mpsz":
            for t in stack:
                if t == 0:
                    idx = {"m": 34, "p": 35, "s": 36}[ch]
                else:
                    idx = "mpsz".index(ch) * 9 + t - 1
                ret[idx] += 1
            stack = []
        elif ch in " \t\n":
            pass
        else:
            raise ValueError(ch)
    return ret


def hand(s):
    h = hand_with_aka(s)
    ret = h[:34].copy()
    ret[4] += h[34]
    ret[13] += h[35]
    ret[22] += h[36]
    return ret


def tile37_to_list(t37):
    out = []
    for tid, c in enumerate(t37):
        if tid < 34:
            out += [tid] * int(c)
        elif c:
            out.append(tid)
    return out


# ---------------------------------------------------------------- events
def pack_event(ev):
    """mjai dict -> int32[EV_INTS]."""
    p = np.zeros(EV_INTS, dtype=np.int32)
    p[3] = 37
    p[4:8] = 37
    p[19:71] = 37
    p[76] = -1
    p[77:82] = 37
    t = ev["type"]
    p[0] = EV_ID[t]
    if "actor" in ev:
        p[1] = ev["actor"]
    if "target" in ev:
        p[2] = ev["target"]
    if "pai" in ev:
        p[3] = TILE_ID[ev["pai"]]
    if "consumed" in ev:
        for i, c in enumerate(ev["consumed"]):
            p[4 + i] = TILE_ID[c]
    if "tsumogiri" in ev:
        p[8] = int(ev["tsumogiri"])
    if t == "start_kyoku":
        p[9] = TILE_ID[ev["bakaze"]]
        p[10] = TILE_ID[ev["dora_marker"]]
        p[11] = ev["kyoku"]
        p[12] = ev["honba"]
        p[13] = ev["kyotaku"]
        p[14] = ev["oya"]
        p[15:19] = ev["scores"]
        for s in range(4):
            for i, tile in enumerate(ev["tehais"][s]):
                p[19 + s * 13 + i] = TILE_ID[tile]
    if t == "dora":
        p[10] = TILE_ID[ev["dora_marker"]]
    if ev.get("deltas") is not None:
        p[71] = 1
        p[72:76] = ev["deltas"]
    if ev.get("ura_markers") is not None:
        p[76] = len(ev["ura_markers"])
        for i, u in enumerate(ev["ura_markers"]):
            p[77 + i] = TILE_ID[u]
    return p


def unpack_event(p):
    """int32[EV_INTS] -> mjai dict (field order follows mjai/event.rs:20-120 serialisation)."""
    t = EV_TYPES[p[0]]
    ev = {"type": t}
    tn = lambda i: TILE_NAMES[i]
    if t == "start_kyoku":
        ev.update(bakaze=tn(p[9]), dora_marker=tn(p[10]), kyoku=int(p[11]), honba=int(p[12]), kyotaku=int(p[13]),
                  oya=int(p[14]), scores=[int(x) for x in p[15:19]],
                  tehais=[[tn(p[19 + s * 13 + i]) for i in range(13)] for s in range(4)])
    elif t == "tsumo":
        ev.update(actor=int(p[1]), pai=tn(p[3]))
    elif t == "dahai":
        ev.update(actor=int(p[1]), pai=tn(p[3]), tsumogiri=bool(p[8]))
    elif t in ("chi", "pon"):
        ev.update(actor=int(p[1]), target=int(p[2]), pai=tn(p[3]), consumed=[tn(p[4]), tn(p[5])])
    elif t == "daiminkan":
        ev.update(actor=int(p[1]), target=int(p[2]), pai=tn(p[3]), consumed=[tn(p[4]), tn(p[5]), tn(p[6])])
    elif t == "kakan":
        ev.update(actor=int(p[1]), pai=tn(p[3]), consumed=[tn(p[4]), tn(p[5]), tn(p[6])])
    elif t == "ankan":
        ev.update(actor=int(p[1]), consumed=[tn(p[4]), tn(p[5]), tn(p[6]), tn(p[7])])
    elif t == "dora":
        ev.update(dora_marker=tn(p[10]))
    elif t in ("reach", "reach_accepted"):
        ev.update(actor=int(p[1]))
    elif t == "hora":
        ev.update(actor=int(p[1]), target=int(p[2]))
        if p[71]:
            ev["deltas"] = [int(x) for x in p[72:76]]
        if p[76] >= 0:
            ev["ura_markers"] = [tn(p[77 + i]) for i in range(p[76])]
    elif t == "ryukyoku":
        if p[71]:
            ev["deltas"] = [int(x) for x in p[72:76]]
    return ev


class PlayerState:
    """Handle on an oracle PlayerState (reference: state/player_state.rs:142-167)."""

    def __init__(self, player_id=0, _handle=None, _own=True):
        self.h = _handle if _handle is not None else lib().mjo_ps_new(player_id)
        self._own = _own

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().mjo_ps_free(self.h)
            self.h = None

    def clone(self):
        return PlayerState(_handle=lib().mjo_ps_clone(self.h))

    def update(self, ev):
        if isinstance(ev, str):
            ev = json.loads(ev)
        p = pack_event(ev)
        cans = np.zeros(13, dtype=np.int32)
        _check(lib().mjo_ps_update(self.h, ptr(p), ptr(cans)))
        return dict(zip(CANS, (int(x) for x in cans)))

    def validate_reaction(self, ev):
        if isinstance(ev, str):
            ev = json.loads(ev)
        p = pack_event(ev)
        _check(lib().mjo_ps_validate_reaction(self.h, ptr(p)))

    def encode_obs(self, version, at_kan_select=False):
        rows = {1: 938, 2: 942, 3: 934, 4: 1012}[version]
        obs = np.empty((rows, 34), dtype=np.float32)
        mask = np.zeros(46, dtype=np.uint8)
        _check(lib().mjo_ps_encode_obs(self.h, version, int(at_kan_select), ptr(obs), ptr(mask)))
        return obs, mask.astype(bool)

    def set_tehai(self, tehai34, len_div3):
        t = np.ascontiguousarray(tehai34, dtype=np.uint8)
        lib().mjo_ps_set_tehai(self.h, ptr(t), len_div3)

    def call(self, what, arg=0):
        return _check(lib().mjo_ps_call(self.h, what, arg))

    def agari_points(self, is_ron, ura=()):
        u = np.array([TILE_ID[x] if isinstance(x, str) else x for x in ura], dtype=np.uint8)
        out = np.zeros(3, dtype=np.int32)
        _check(lib().mjo_ps_agari_points(self.h, int(is_ron), ptr(u), len(u), ptr(out)))
        return dict(ron=int(out[0]), tsumo_ko=int(out[1]), tsumo_oya=int(out[2]))

    def snapshot(self):
        o = np.zeros(256, dtype=np.int32)
        lib().mjo_ps_snapshot(self.h, ptr(o))
        return dict(
            shanten=int(o[0]), at_furiten=bool(o[1]), has_next_shanten_discard=bool(o[2]), tehai_len_div3=int(o[3]),
            is_menzen=bool(o[4]), tiles_left=int(o[5]), at_turn=int(o[6]), rank=int(o[7]), doras_seen=int(o[8]),
            kans_on_board=int(o[9]), doras_owned=[int(x) for x in o[10:14]], waits=o[14:48].astype(bool),
            tehai=o[48:82].copy(), tiles_seen=o[82:116].copy(), keep_shanten_discards=o[116:150].astype(bool),
            next_shanten_discards=o[150:184].astype(bool), forbidden_tiles=o[184:218].astype(bool),
            cans=dict(zip(CANS, (int(x) for x in o[218:231]))), akas_in_hand=o[231:234].astype(bool),
            real_time_shanten=int(o[234]), can_w_riichi=bool(o[235]), at_ippatsu=bool(o[236]),
            at_rinshan=bool(o[237]), scores=[int(x) for x in o[238:242]], n_ankan_cand=int(o[242]), n_kakan_cand=int(o[243]),
            last_self_tsumo=int(o[244]),
        )

    def set_scores(self, scores):
        a = np.array(scores, dtype=np.int32)
        lib().mjo_ps_set_scores(self.h, ptr(a))

    def get_rank(self, scores_rel):
        a = np.array(scores_rel, dtype=np.int32)
        return lib().mjo_ps_get_rank(self.h, ptr(a))

    def kawa(self, rel):
        out = np.zeros(64, dtype=np.uint64)
        n = lib().mjo_ps_kawa(self.h, rel, ptr(out), 64)
        return out[:n]

    def uncond_tenpai(self):
        out = np.zeros(34, dtype=np.uint8)
        _check(lib().mjo_ps_uncond_tenpai(self.h, ptr(out)))
        return out.astype(bool)

    def sp_tables(self):
        f = np.zeros((14, 51), dtype=np.float32)
        i = np.zeros((14, 73), dtype=np.int32)
        n = _check(lib().mjo_ps_sp_tables(self.h, ptr(f), ptr(i), 14))
        return _sp_unpack(f, i, n)

    def scene(self, enable_quick_eval=True):
        out = np.zeros(5, dtype=np.int32)
        _check(lib().mjo_ps_scene(self.h, int(enable_quick_eval), ptr(out)))
        return dict(can_act=bool(out[0]), quick_eval=bool(out[1]), quick_pai=int(out[2]), quick_tsumogiri=bool(out[3]),
                    need_kan_select=bool(out[4]))

    def decode_action(self, action, kan_tile=-1):
        out = np.zeros(EV_INTS, dtype=np.int32)
        _check(lib().mjo_ps_decode_action(self.h, action, kan_tile, ptr(out)))
        return unpack_event(out)


def _sp_unpack(f, i, n):
    out = []
    for k in range(min(n, len(f))):
        m = int(i[k, 1])
        nr = int(i[k, 4])
        out.append(dict(
            tile=int(i[k, 0]), tenpai_probs=f[k, :m].copy(), win_probs=f[k, 17:17 + m].copy(),
            exp_values=f[k, 34:34 + m].copy(), shanten_down=bool(i[k, 2]), num_required_tiles=int(i[k, 3]),
            required_tiles=[(int(i[k, 5 + 2 * j]), int(i[k, 6 + 2 * j])) for j in range(nr)],
        ))
    return out


def sp_calc(tehai, tiles_seen, *, len_div3=4, is_menzen=True, bakaze=27, jikaze=27, num_doras_in_fuuro=0,
            calc_double_riichi=False, calc_haitei=False, prefer_riichi=True, sort_result=True, maximize_win_prob=False,
            calc_tegawari=False, calc_shanten_down=False, can_discard=True, tsumos_left=17, cur_shanten=None,
            dora_indicators=(), akas_in_hand=(0, 0, 0), akas_seen=(0, 0, 0), chis=(), pons=(), minkans=(), ankans=()):
    cfg = np.zeros(41, dtype=np.int32)
    cfg[:15] = [len_div3, is_menzen, bakaze, jikaze, num_doras_in_fuuro, calc_double_riichi, calc_haitei,
                prefer_riichi, sort_result, maximize_win_prob, calc_tegawari, calc_shanten_down, can_discard,
                tsumos_left, cur_shanten]
    cfg[15] = len(dora_indicators)
    for k, d in enumerate(dora_indicators):
        cfg[16 + k] = d
    for j, m in enumerate((chis, pons, minkans, ankans)):
        cfg[21 + j] = len(m)
        for k, t in enumerate(m):
            cfg[25 + 4 * j + k] = t
    t = np.ascontiguousarray(tehai, dtype=np.uint8)
    ts = np.ascontiguousarray(tiles_seen, dtype=np.uint8)
    ah = np.array(akas_in_hand, dtype=np.uint8)
    asn = np.array(akas_seen, dtype=np.uint8)
    f = np.zeros((14, 51), dtype=np.float32)
    i = np.zeros((14, 73), dtype=np.int32)
    n = _check(lib().mjo_sp_calc(ptr(cfg), ptr(t), ptr(ah), ptr(ts), ptr(asn), ptr(f), ptr(i), 14))
    return _sp_unpack(f, i, n)


def calc_shanten(tehai34, len_div3, which=0):
    t = np.ascontiguousarray(tehai34, dtype=np.uint8)
    return lib().mjo_calc_shanten(ptr(t), len_div3, which)


def agari(tehai34, winning_tile, is_ron, *, chis=(), pons=(), minkans=(), ankans=(), bakaze=27, jikaze=27, mode=1,
          additional_hans=0, doras=0):
    t = np.ascontiguousarray(tehai34, dtype=np.uint8)
    melds = np.zeros(16, dtype=np.uint8)
    n = np.zeros(4, dtype=np.int32)
    for j, m in enumerate((chis, pons, minkans, ankans)):
        n[j] = len(m)
        melds[4 * j:4 * j + len(m)] = list(m)
    out = np.zeros(3, dtype=np.int32)
    _check(lib().mjo_agari(ptr(t), ptr(melds), ptr(n), bakaze, jikaze, winning_tile, int(is_ron), mode,
                           additional_hans, doras, ptr(out)))
    if mode == 2:
        return bool(out[0])
    if out[0] == 0:
        return None
    if out[0] == 2:
        return ("yakuman", int(out[2]))
    return ("normal", int(out[1]), int(out[2]))


def deal(nonce, key, kyoku, honba, algo=0):
    seq = np.zeros(136, dtype=np.uint8)
    lib().mjo_deal(nonce, key, kyoku, honba, algo, ptr(seq))
    return seq


class Arena:
    """Lock-step batch of hanchan (reference: arena/game.rs:230-316)."""

    def __init__(self, seeds, deal_algo=0, enable_quick_eval=True, version=4, keep_log=True):
        self.n = len(seeds)
        nonces = np.array([s[0] for s in seeds], dtype=np.uint64)
        keys = np.array([s[1] for s in seeds], dtype=np.uint64)
        self.version = version
        self.h = lib().mjo_arena_new(self.n, ptr(nonces), ptr(keys), deal_algo, int(enable_quick_eval), version,
                                     int(keep_log))

    def __del__(self):
        if getattr(self, "h", None):
            lib().mjo_arena_free(self.h)
            self.h = None

    def restart(self, g, nonce):
        """Finished slot g starts a fresh hanchan on (nonce, same key) — the oracle side of the pool's refill mode."""
        _check(lib().mjo_arena_restart(self.h, int(g), int(nonce)))

    def park(self):
        """Every slot finished before the first cycle (staggered first start): `restart` then brings slot t into play at its cycle."""
        _check(lib().mjo_arena_park(self.h))

    def poll(self):
        n = _check(lib().mjo_arena_poll(self.h))
        rows = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().mjo_arena_rows(self.h, ptr(rows))
        return rows

    def encode(self, row0, row1, want_obs=True, threads=0):
        """threads > 1: rows are independent and the encoder is const, so disjoint row ranges are encoded concurrently
        (ctypes releases the GIL) — keeps the big-pool parity tests inside their time budget."""
        n = row1 - row0
        masks = np.zeros((n, 46), dtype=np.uint8)
        obs = None
        if want_obs:
            rows = {1: 938, 2: 942, 3: 934, 4: 1012}[self.version]
            obs = np.empty((n, rows, 34), dtype=np.float32)
        if threads > 1 and n >= 4 * threads:
            from concurrent.futures import ThreadPoolExecutor

            cuts = [row0 + (n * k) // (4 * threads) for k in range(4 * threads + 1)]

            def part(k):
                a, b = cuts[k], cuts[k + 1]
                return lib().mjo_arena_encode(self.h, a, b, ptr(obs[a - row0:]) if want_obs else None, ptr(masks[a - row0:]))

            with ThreadPoolExecutor(threads) as ex:
                for rc in ex.map(part, range(4 * threads)):
                    _check(rc)
            return obs, masks
        _check(lib().mjo_arena_encode(self.h, row0, row1, ptr(obs) if want_obs else None, ptr(masks)))
        return obs, masks

    def encode_oracle(self, row0, row1, version):
        """Invisible obs (board.rs:679-782) of rows [row0, row1): f32 [n, 211|217, 34]."""
        rows = 211 if version == 1 else 217
        out = np.empty((row1 - row0, rows, 34), dtype=np.float32)
        _check(lib().mjo_arena_encode_oracle(self.h, row0, row1, version, ptr(out)))
        return out

    def commit(self, actions, q_values=None):
        """q_values (f32 [rows, 46]) switches the rule-based agari guard on for every seat (agent/mortal.rs:319-336)."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if q_values is None:
            return _check(lib().mjo_arena_commit(self.h, ptr(a)))
        q = np.ascontiguousarray(q_values, dtype=np.float32)
        assert q.shape == (len(a), 46)
        return _check(lib().mjo_arena_commit_q(self.h, ptr(a), ptr(q)))

    @property
    def guard_hits(self):
        return lib().mjo_arena_guard_hits(self.h)

    @property
    def n_live(self):
        return lib().mjo_arena_n_live(self.h)

    @property
    def steps(self):
        return lib().mjo_arena_steps(self.h)

    def done_flags(self):
        """done flag of every slot (one call; `result(g)` per slot costs microseconds each on a 16 k-table pool, every cycle)."""
        out = np.zeros(self.n, dtype=np.uint8)
        lib().mjo_arena_done_flags(self.h, ptr(out))
        return out

    def result(self, g):
        s = np.zeros(4, dtype=np.int32)
        d = C.c_int(0)
        lib().mjo_arena_result(self.h, g, ptr(s), C.byref(d))
        return s, bool(d.value)

    def game_view(self, g):
        o = np.zeros(16, dtype=np.int32)
        lib().mjo_arena_game_view(self.h, g, ptr(o))
        return o

    def player_state(self, g, seat):
        h = lib().mjo_arena_player_state(self.h, g, seat)
        return PlayerState(_handle=h, _own=False) if h else None

    def log(self, g):
        n = lib().mjo_arena_log(self.h, g, None, 0)
        buf = np.zeros((n, EV_INTS), dtype=np.int32)
        lib().mjo_arena_log(self.h, g, ptr(buf), n)
        return [unpack_event(buf[i]) for i in range(n)]


def random_actions(masks, rows, cycle, seed=0x9E3779B97F4A7C15):
    """Counter-based uniform-random legal action per row, keyed by (game, seat, is_kan, cycle).

    Shared by the oracle and the HIP path so both consume identical action streams.
    """
    masks = np.asarray(masks, dtype=bool)
    n = len(masks)
    if n == 0:
        return np.zeros(0, dtype=np.int32)
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) ^ (rows[:, 0] * np.uint64(0xD1B54A32D192ED03))
             ^ (rows[:, 1] * np.uint64(0x8CB92BA72F3D8DD7)) ^ (rows[:, 2] * np.uint64(0xAEF17502108EF2D9))
             ^ (np.array([cycle], dtype=np.uint64) * np.uint64(0x94D049BB133111EB)))
        # splitmix64 finaliser
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    cnt = masks.sum(axis=1).astype(np.uint64)
    assert (cnt > 0).all(), "row with empty mask"
    k = (x >> np.uint64(33)) % cnt
    cs = np.cumsum(masks, axis=1)
    act = (cs > k[:, None].astype(np.int64)).argmax(axis=1)
    return act.astype(np.int32)
