"""libriichi.state.PlayerState (state/player_state.rs:142-167, pyo3 surface) on the device path.

One table of a TablePool plays the role of the reference's per-player state machine: `update(event)` applies an mjai
event through the same HIP event handlers the arena uses (`mj_table_apply_event` -> mj_rules.h ev_*), `encode_obs` runs the
arena's snapshot + encode kernels for this seat, the getters read the table record back.  Meant for tests and debugging
— the reference's own scenario tests (state/test.rs) run against it in tests/test_gpu_state.py.

Hidden information is written the reference's way: other seats' haipai and draws are "?" tiles.
"""
import json

import numpy as np

from . import mjai_log
from ._lib import MortalAmdError, check, lib
from .pool import TablePool, _stream

_CAN_BITS = ["can_discard", "can_chi_low", "can_chi_mid", "can_chi_high", "can_pon", "can_daiminkan", "can_kakan",
             "can_ankan", "can_riichi", "can_tsumo_agari", "can_ron_agari", "can_ryukyoku"]
Q_AGARI_POINTS, Q_RULE_BASED_AGARI, Q_REAL_TIME_SHANTEN, Q_DORAS_OWNED, Q_ADD_DORA, Q_SET_SCORES = range(6)
Q_VALIDATE_REACTION = 8


class ActionCandidate:
    """state/action.rs:13-89."""

    def __init__(self, bits=0, target_actor=0):
        for i, name in enumerate(_CAN_BITS):
            setattr(self, name, bool((bits >> i) & 1))
        self.target_actor = int(target_actor)

    can_chi = property(lambda s: s.can_chi_low or s.can_chi_mid or s.can_chi_high)
    can_kan = property(lambda s: s.can_daiminkan or s.can_kakan or s.can_ankan)
    can_agari = property(lambda s: s.can_tsumo_agari or s.can_ron_agari)
    can_pass = property(lambda s: s.can_chi or s.can_pon or s.can_daiminkan or s.can_ron_agari)
    can_act = property(lambda s: s.can_discard or s.can_chi or s.can_pon or s.can_kan or s.can_riichi or s.can_agari
                       or s.can_ryukyoku)

    def __repr__(self):
        on = [n for n in _CAN_BITS if getattr(self, n)]
        return f"ActionCandidate({', '.join(on)}; target_actor={self.target_actor})"


class PlayerState:
    pool_cls = TablePool  # the test suite substitutes the host emulation of the same kernels (tests/host/emu_pool.py)

    def __init__(self, player_id, device="cuda:0"):
        if not 0 <= int(player_id) <= 3:
            raise ValueError("player_id must be within 0..3")
        self.player_id = int(player_id)
        self._pool = self.pool_cls(1, version=4, device=device)
        self._pool.reset([(0, 0)])
        self._tbl = 0
        self._owns = True
        self._cache = None

    @classmethod
    def view(cls, pool, table, player_id):
        """A read-only copy of seat `player_id`'s state on table `table` of an arena pool, taken now — what the reference
        hands to an mjai-log engine as `GameState.state` (`state.clone()`, agent/mjai_log.rs:104-118).  The getters and
        `brief_info()` answer from the copy; `update` / `encode_obs` / device queries are not available on it."""
        self = cls.__new__(cls)
        self.player_id = int(player_id)
        self._pool = None
        self._tbl = int(table)
        self._owns = False
        self._cache = pool.debug_table(int(table))
        return self

    def close(self):
        if self._pool is not None and self._owns:
            self._pool.close()
        self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- update (state/update.rs:23-39)
    def update(self, mjai_json):
        ev = json.loads(mjai_json) if isinstance(mjai_json, str) else mjai_json
        if ev["type"] in ("start_game", "end_game"):
            ev = {"type": "end_kyoku"}  # same effect on a PlayerState: only the per-event reset of last_cans
        words = mjai_log.encode_events([ev])
        self._need_pool()
        check(self._pool._L.mj_table_apply_event(self._pool.h, 0, words.ctypes.data, len(words), self._pool._stream()))
        self._cache = None
        code, _ = self._pool.first_error()
        if code:
            raise MortalAmdError(f"rule violation while applying {ev} (error code {code})")
        return self.last_cans

    def reaction_accepted_by_device(self, mjai_json):
        """The step kernel's own verdict on an explicit reaction (what a raw `mj_step_ev` caller gets): True = it would be applied,
        False = the table would go into MJ_ERR_ILLEGAL_ACTION.  validate_reaction's checks plus: a call / ron must name the seat
        that discarded (`last_cans.target_actor`)."""
        from .arena import pack_reaction

        ev = json.loads(mjai_json) if isinstance(mjai_json, str) else mjai_json
        w = pack_reaction(ev)
        out = self._query(Q_VALIDATE_REACTION, [int(np.int32(np.uint32(w & 0xFFFFFFFF))), int(np.int32(np.uint32(w >> 32)))])
        return int(out[0]) == 0

    # ---- validate_reaction (state/action.rs:91-228)
    def validate_reaction(self, mjai_json):
        """Raises ValueError when `action` is not a valid reaction to the current state (same checks and messages)."""
        ev = json.loads(mjai_json) if isinstance(mjai_json, str) else mjai_json
        cans = self.last_cans
        t = ev["type"]

        def ensure(cond, msg):
            if not cond:
                raise ValueError(msg)

        if t == "ryukyoku":
            return ensure(cans.can_ryukyoku, "cannot ryukyoku")
        if t == "none":
            return None
        ensure("actor" in ev, "action does not have actor and is not ryukyoku")
        ensure(ev["actor"] == self.player_id, f"actor is {ev['actor']}, not self ({self.player_id})")
        tid = mjai_log.TILE_ID
        deaka = lambda x: {34: 4, 35: 13, 36: 22}.get(x, x)
        hand, akas = self._hand(), self.akas_in_hand
        tbl = self._table()
        raw = int(tbl["last_kawa_tile"][self.player_id])
        last_kawa = None if raw >= 38 else raw
        raw = int(tbl["last_self_tsumo"][self.player_id])
        last_tsumo = None if raw >= 38 else raw

        def in_hand(tiles):
            for name in tiles:
                x = tid[name]
                ensure(hand[deaka(x)] > 0, f"{name} is not in hand")
                if x >= 34:
                    ensure(akas[x - 34], f"{name} is not in hand")

        if t == "dahai":
            ensure(cans.can_discard, "cannot discard")
            in_hand([ev["pai"]])
            if ev["tsumogiri"]:
                ensure(last_tsumo is not None, "tsumogiri but the player has not dealt any tile yet")
                ensure(last_tsumo == tid[ev["pai"]], "cannot tsumogiri")
        elif t == "reach":
            ensure(cans.can_riichi, "cannot riichi")
        elif t == "chi":
            ensure((ev["target"] + 1) % 4 == ev["actor"], "chi from non-kamicha")
            ensure(last_kawa is not None and last_kawa == tid[ev["pai"]], "chi target is not the last kawa tile")
            in_hand(ev["consumed"])
            a, b = sorted(deaka(tid[x]) for x in ev["consumed"])
            pai = deaka(tid[ev["pai"]])
            if pai < a:
                ensure(cans.can_chi_low, "cannot chi low")
            elif pai < b:
                ensure(cans.can_chi_mid, "cannot chi mid")
            else:
                ensure(cans.can_chi_high, "cannot chi high")
        elif t in ("pon", "daiminkan"):
            ensure(ev["target"] != ev["actor"], f"{t} from itself")
            ensure(last_kawa is not None and last_kawa == tid[ev["pai"]], f"{t} target is not the last kawa tile")
            ensure(cans.can_pon if t == "pon" else cans.can_daiminkan, f"cannot {t}")
            in_hand(ev["consumed"])
        elif t == "kakan":
            ensure(cans.can_kakan, "cannot kakan")
            ensure(deaka(tid[ev["pai"]]) in self._kakan_cand, f"cannot kakan {ev['pai']}")
            in_hand([ev["pai"]])
        elif t == "ankan":
            ensure(cans.can_ankan, "cannot ankan")
            tile = deaka(tid[ev["consumed"][0]])
            ensure(tile in self._ankan_cand, f"cannot ankan {mjai_log.TILE_NAMES[tile]}")
            in_hand(ev["consumed"])
        elif t == "hora":
            if ev["target"] == self.player_id:
                ensure(cans.can_tsumo_agari, "cannot tsumo agari")
            else:
                ensure(cans.can_ron_agari, "cannot ron agari")
        else:
            raise ValueError(f"unexpected action {ev!r}")
        return None

    def _need_pool(self):
        if self._pool is None:
            raise MortalAmdError("this PlayerState is a read-only copy of an arena table (PlayerState.view)")

    def _table(self):
        if self._cache is None:
            self._need_pool()
            self._cache = self._pool.debug_table(0)
        return self._cache

    def _query(self, what, args=()):
        self._need_pool()
        a = np.zeros(8, dtype=np.int32)
        a[:len(args)] = args
        out = np.zeros(8, dtype=np.int32)
        check(self._pool._L.mj_table_query(self._pool.h, 0, self.player_id, what, a.ctypes.data, out.ctypes.data, self._pool._stream()))
        self._cache = None
        return out

    # ---- obs (state/obs_repr.rs:776-791)
    def encode_obs(self, version, at_kan_select):
        self._need_pool()
        self._pool.configure(0, version=int(version))
        check(self._pool._L.mj_table_mark_row(self._pool.h, 0, self.player_id, int(bool(at_kan_select)), self._pool._stream()))
        import ctypes as C

        out = (C.c_int32 * 2)()
        check(self._pool._L.mj_rows_count(self._pool.h, out, self._pool._stream()))
        self._pool.n_rows = [out[0], out[1]]
        obs, masks = self._pool.encode(0)
        return obs[0].cpu().numpy(), masks[0].cpu().numpy()

    # ---- getters (state/getter.rs)
    @property
    def last_cans(self):
        t = self._table()
        return ActionCandidate(int(t["cans"][self.player_id]), int(t["cans_target"][self.player_id]))

    def _hand(self):
        t = self._table()
        mp, sz = int(t["hand_mp"][self.player_id]), int(t["hand_sz"][self.player_id])
        return [(mp >> (3 * i)) & 7 for i in range(18)] + [(sz >> (3 * i)) & 7 for i in range(16)]

    def _bits34(self, name):
        v = int(self._table()[name][self.player_id])
        return [bool((v >> i) & 1) for i in range(34)]

    tehai = property(lambda s: s._hand())
    waits = property(lambda s: s._bits34("waits"))
    shanten = property(lambda s: int(np.int8(s._table()["shanten"][s.player_id])))
    at_furiten = property(lambda s: bool(int(s._table()["pflags"][s.player_id]) & (1 << 5)))
    has_next_shanten_discard = property(lambda s: bool(s._table()["has_next_shanten"][s.player_id]))
    at_turn = property(lambda s: int(s._table()["at_turn"][s.player_id]))
    tiles_left = property(lambda s: int(s._table()["tiles_left"][0]))
    kyotaku = property(lambda s: int(s._table()["kyotaku"][0]))
    honba = property(lambda s: int(s._table()["honba"][0]))
    is_menzen = property(lambda s: bool(int(s._table()["pflags"][s.player_id]) & (1 << 7)))
    akas_in_hand = property(lambda s: [bool((int(s._table()["akas_in_hand"][s.player_id]) >> i) & 1) for i in range(3)])
    _ankan_cand = property(lambda s: [i for i, b in enumerate(s._bits34("ankan_cand")) if b])
    _kakan_cand = property(lambda s: [i for i, b in enumerate(s._bits34("kakan_cand")) if b])
    # the rest of the pyo3 surface (state/getter.rs:8-156)
    kyoku = property(lambda s: int(s._table()["kyoku"][0]) & 3)          # within the round (update.rs:151-153)
    is_oya = property(lambda s: (int(s._table()["kyoku"][0]) & 3) == s.player_id)
    can_w_riichi = property(lambda s: bool(int(s._table()["pflags"][s.player_id]) & (1 << 1)))
    self_riichi_declared = property(lambda s: bool((int(s._table()["riichi_declared"][0]) >> s.player_id) & 1))
    self_riichi_accepted = property(lambda s: bool((int(s._table()["riichi_accepted"][0]) >> s.player_id) & 1))

    def _melds(self, name, kind):
        t = self._table()
        n = int(t["n_melds"].reshape(4, 4)[self.player_id][kind])
        return [int(x) for x in t[name].reshape(4, 4)[self.player_id][:n]]

    chis = property(lambda s: s._melds("chis", 0))
    pons = property(lambda s: s._melds("pons", 1))
    minkans = property(lambda s: s._melds("minkans", 2))
    ankans = property(lambda s: s._melds("ankans", 3))

    def _tile_or_none(self, name):
        raw = int(self._table()[name][self.player_id])
        return None if raw >= 38 else mjai_log.TILE_NAMES[raw]

    def last_self_tsumo(self):
        return self._tile_or_none("last_self_tsumo")

    def last_kawa_tile(self):
        return self._tile_or_none("last_kawa_tile")

    def ankan_candidates(self):
        return [mjai_log.TILE_NAMES[t] for t in self._ankan_cand]

    def kakan_candidates(self):
        return [mjai_log.TILE_NAMES[t] for t in self._kakan_cand]

    @property
    def scores(self):
        """Relative to the player: index 0 is self (player_state.rs:40)."""
        sc = [int(x) for x in self._table()["scores"]]
        return sc[self.player_id:] + sc[:self.player_id]

    @property
    def doras_owned(self):
        return [int(x) for x in self._query(Q_DORAS_OWNED)[:4]]

    def real_time_shanten(self):
        return int(self._query(Q_REAL_TIME_SHANTEN)[0])

    def rule_based_agari(self):
        return bool(self._query(Q_RULE_BASED_AGARI)[0])

    def agari_points(self, is_ron, ura_indicators=()):
        ura = [mjai_log.TILE_ID[x] if isinstance(x, str) else int(x) for x in ura_indicators]
        out = self._query(Q_AGARI_POINTS, [int(bool(is_ron)), len(ura)] + ura + [0] * (5 - len(ura)))
        if not out[0]:
            raise MortalAmdError("cannot agari / not a hora hand")
        return dict(ron=int(out[1]), tsumo_ko=int(out[2]), tsumo_oya=int(out[3]))

    def get_rank(self, scores_rel):
        """update.rs:966-972 + rankings.rs:8-21: rank of self for scores given relative to self."""
        abs_scores = [0] * 4
        for i in range(4):
            abs_scores[(i + self.player_id) % 4] = scores_rel[i]
        order = sorted(range(4), key=lambda i: -abs_scores[i])
        return order.index(self.player_id)

    def discard_candidates_with_unconditional_tenpai(self):
        """agent_helper.rs:88-197, read from the obs plane that carries it (v4 row 877)."""
        obs, _ = self.encode_obs(4, False)
        return [bool(x) for x in obs[877]]

    # test hooks (state/test.rs pokes these fields directly)
    def set_scores_rel(self, scores_rel):
        abs_scores = [0] * 4
        for i in range(4):
            abs_scores[(i + self.player_id) % 4] = int(scores_rel[i])
        self._query(Q_SET_SCORES, abs_scores)

    def add_dora_indicator(self, tile):
        self._query(Q_ADD_DORA, [mjai_log.TILE_ID[tile] if isinstance(tile, str) else int(tile)])

    def brief_info(self):
        names = mjai_log.TILE_NAMES
        h = self._hand()
        hand = " ".join(names[t] for t in range(34) for _ in range(h[t]))
        return (f"player (abs): {self.player_id}\nscores (rel): {self.scores}\ntehai: {hand}\nshanten: {self.shanten}\n"
                f"furiten: {self.at_furiten}\ntiles left: {self.tiles_left}\nlast cans: {self.last_cans!r}")
