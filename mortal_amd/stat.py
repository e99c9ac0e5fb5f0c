"""libriichi.stat.Stat (stat.rs): statistics of one player over a directory of mjai game logs.

A pure event-stream reduction (stat.rs:263-441 `from_game`): no game-state replay is needed, so this lives on the host.
Counter names, update rules, derived-rate getters and the text report follow the reference one to one
(fields stat.rs:30-126, getters :500-779, Display :128-256).
"""
import glob
import gzip
import json
import math
import os

_FIELDS = [
    "game", "round", "oya", "point", "rank_1", "rank_2", "rank_3", "rank_4", "tobi",
    "fuuro", "fuuro_num", "fuuro_point", "fuuro_agari", "fuuro_agari_jun", "fuuro_agari_point", "fuuro_houjuu",
    "agari", "agari_as_oya", "agari_jun", "agari_point_oya", "agari_point_ko",
    "houjuu", "houjuu_jun", "houjuu_to_oya", "houjuu_point_to_oya", "houjuu_point_to_ko",
    "riichi", "riichi_as_oya", "riichi_jun", "riichi_agari", "riichi_agari_point", "riichi_agari_jun", "riichi_houjuu",
    "riichi_ryukyoku", "riichi_point", "chasing_riichi", "riichi_got_chased",
    "dama_agari", "dama_agari_jun", "dama_agari_point", "ryukyoku", "ryukyoku_point", "yakuman", "nagashi_mangan",
]


def _div(a, b):
    """f64 division with Rust semantics (x/0 = ±inf, 0/0 = NaN)."""
    a, b = float(a), float(b)
    if b == 0.0:
        return math.nan if a == 0.0 else math.copysign(math.inf, a)
    return a / b


class Stat:
    __slots__ = _FIELDS

    def __init__(self, **kw):
        for f in _FIELDS:
            setattr(self, f, int(kw.get(f, 0)))

    # ---- derive_more Add / AddAssign / Sum
    def __add__(self, other):
        return Stat(**{f: getattr(self, f) + getattr(other, f) for f in _FIELDS})

    def __iadd__(self, other):
        for f in _FIELDS:
            setattr(self, f, getattr(self, f) + getattr(other, f))
        return self

    def __radd__(self, other):
        return self if other == 0 else self.__add__(other)

    def __eq__(self, other):
        return isinstance(other, Stat) and all(getattr(self, f) == getattr(other, f) for f in _FIELDS)

    # ---- stat.rs:263-441
    @staticmethod
    def from_game(events, player_id):
        st = Stat(game=1)
        cur = [0, 0, 0, 0]
        riichi_declared = riichi_accepted = others_riichi_declared = False
        cur_oya = jun = fuuro_num = 0
        for ev in events:
            t = ev["type"]
            if t == "start_kyoku":
                st.round += 1
                cur = list(ev["scores"])
                riichi_declared = riichi_accepted = others_riichi_declared = False
                cur_oya = ev["oya"]
                if cur_oya == player_id:
                    st.oya += 1
                jun = fuuro_num = 0
            elif t == "dahai":
                if ev["actor"] == player_id:
                    jun += 1
            elif t in ("chi", "pon", "daiminkan"):
                if ev["actor"] == player_id:
                    fuuro_num += 1
            elif t == "reach":
                if ev["actor"] == player_id:
                    riichi_declared = True
                    st.riichi += 1
                    st.riichi_jun += jun
                    if cur_oya == player_id:
                        st.riichi_as_oya += 1
                    if others_riichi_declared:
                        st.chasing_riichi += 1
                elif riichi_declared:
                    st.riichi_got_chased += 1
                else:
                    others_riichi_declared = True
            elif t == "reach_accepted":
                cur[ev["actor"]] -= 1000
                if ev["actor"] == player_id:
                    riichi_accepted = True
            elif t == "hora":
                deltas = ev.get("deltas")
                if deltas is None:
                    raise ValueError("deltas is required for analyzing")
                cur = [a + b for a, b in zip(cur, deltas)]
                actor, target = ev["actor"], ev["target"]
                if actor == player_id:
                    point = deltas[player_id] - (1000 if riichi_accepted else 0)
                    st.agari += 1
                    st.agari_jun += jun
                    if cur_oya == player_id:
                        st.agari_as_oya += 1
                        st.agari_point_oya += point
                    else:
                        st.agari_point_ko += point
                    if riichi_accepted:
                        st.riichi_agari += 1
                        st.riichi_agari_jun += jun
                        st.riichi_agari_point += point
                        st.riichi_point += point
                    elif fuuro_num > 0:
                        st.fuuro_agari += 1
                        st.fuuro_agari_jun += jun
                        st.fuuro_agari_point += point
                        st.fuuro_point += point
                    else:
                        st.dama_agari += 1
                        st.dama_agari_jun += jun
                        st.dama_agari_point += point
                    if point >= (48000 if cur_oya == player_id else 32000):  # Point::yakuman(is_oya, 1).ron
                        st.yakuman += 1
                elif target == player_id:
                    point = deltas[player_id]
                    st.houjuu += 1
                    st.houjuu_jun += jun
                    if cur_oya == actor:
                        st.houjuu_to_oya += 1
                        st.houjuu_point_to_oya += point
                    else:
                        st.houjuu_point_to_ko += point
                    if riichi_declared:
                        st.riichi_houjuu += 1
                        st.riichi_point += point
                    elif fuuro_num > 0:
                        st.fuuro_houjuu += 1
                        st.fuuro_point += point
            elif t == "ryukyoku":
                deltas = ev.get("deltas")
                if deltas is None:
                    raise ValueError("deltas is required for analyzing")
                cur = [a + b for a, b in zip(cur, deltas)]
                point = deltas[player_id]
                st.ryukyoku += 1
                st.ryukyoku_point += point
                if riichi_accepted:
                    st.riichi_ryukyoku += 1
                    st.riichi_point += point - 1000
                elif fuuro_num > 0:
                    st.fuuro_point += point
                if point >= 8000:
                    st.nagashi_mangan += 1
            elif t == "end_kyoku":
                if fuuro_num > 0:
                    st.fuuro += 1
                    st.fuuro_num += fuuro_num
        order = sorted(range(4), key=lambda i: -cur[i])  # Rankings::new: stable, ties -> lower seat
        total = sum(cur)
        if total < 100_000:  # assume the sum of scores to be 100k
            cur[order[0]] += 100_000 - total
        final = cur[player_id]
        st.point = final - 25000
        if final < 0:
            st.tobi = 1
        setattr(st, f"rank_{order.index(player_id) + 1}", 1)
        return st

    @staticmethod
    def from_log(log, player_id):
        return Stat.from_game([json.loads(l) for l in log.splitlines() if l.strip()], int(player_id))

    @staticmethod
    def from_dir(dir, player_name, disable_progress_bar=False):
        total = Stat()
        paths = glob.glob(os.path.join(dir, "**", "*.json"), recursive=True)
        paths += glob.glob(os.path.join(dir, "**", "*.json.gz"), recursive=True)
        for path in paths:
            op = gzip.open if path.lower().endswith(".gz") else open
            with op(path, "rt") as f:
                events = [json.loads(l) for l in f.read().splitlines()]
            if not events or events[0].get("type") != "start_game":
                raise ValueError(f"first event is not start_game, got {events[0] if events else None!r}")
            for i, name in enumerate(events[0].get("names", ["", "", "", ""])):
                if name == player_name:
                    total += Stat.from_game(events, i)
        return total

    # ---- stat.rs:500-779
    def total_pt(self, pts):
        return self.rank_1 * pts[0] + self.rank_2 * pts[1] + self.rank_3 * pts[2] + self.rank_4 * pts[3]

    def avg_pt(self, pts):
        return _div(self.total_pt(pts), self.game)

    avg_rank = property(lambda s: s.avg_pt([1, 2, 3, 4]))
    rank_1_rate = property(lambda s: _div(s.rank_1, s.game))
    rank_2_rate = property(lambda s: _div(s.rank_2, s.game))
    rank_3_rate = property(lambda s: _div(s.rank_3, s.game))
    rank_4_rate = property(lambda s: _div(s.rank_4, s.game))
    tobi_rate = property(lambda s: _div(s.tobi, s.game))
    avg_point_per_game = property(lambda s: _div(s.point, s.game))
    avg_point_per_round = property(lambda s: _div(s.point, s.round))
    avg_point_per_agari = property(lambda s: _div(s.agari_point_ko + s.agari_point_oya, s.agari))
    avg_point_per_oya_agari = property(lambda s: _div(s.agari_point_oya, s.agari_as_oya))
    avg_point_per_ko_agari = property(lambda s: _div(s.agari_point_ko, s.agari - s.agari_as_oya))
    avg_point_per_riichi_agari = property(lambda s: _div(s.riichi_agari_point, s.riichi_agari))
    avg_point_per_fuuro_agari = property(lambda s: _div(s.fuuro_agari_point, s.fuuro_agari))
    avg_point_per_dama_agari = property(lambda s: _div(s.dama_agari_point, s.dama_agari))
    avg_point_per_ryukyoku = property(lambda s: _div(s.ryukyoku_point, s.ryukyoku))
    avg_agari_jun = property(lambda s: _div(s.agari_jun, s.agari))
    avg_riichi_agari_jun = property(lambda s: _div(s.riichi_agari_jun, s.riichi_agari))
    avg_fuuro_agari_jun = property(lambda s: _div(s.fuuro_agari_jun, s.fuuro_agari))
    avg_dama_agari_jun = property(lambda s: _div(s.dama_agari_jun, s.dama_agari))
    avg_point_per_houjuu = property(lambda s: _div(s.houjuu_point_to_ko + s.houjuu_point_to_oya, s.houjuu))
    avg_point_per_houjuu_to_oya = property(lambda s: _div(s.houjuu_point_to_oya, s.houjuu_to_oya))
    avg_point_per_houjuu_to_ko = property(lambda s: _div(s.houjuu_point_to_ko, s.houjuu - s.houjuu_to_oya))
    avg_houjuu_jun = property(lambda s: _div(s.houjuu_jun, s.houjuu))
    agari_rate = property(lambda s: _div(s.agari, s.round))
    houjuu_rate = property(lambda s: _div(s.houjuu, s.round))
    riichi_rate = property(lambda s: _div(s.riichi, s.round))
    fuuro_rate = property(lambda s: _div(s.fuuro, s.round))
    ryukyoku_rate = property(lambda s: _div(s.ryukyoku, s.round))
    agari_rate_after_riichi = property(lambda s: _div(s.riichi_agari, s.riichi))
    houjuu_rate_after_riichi = property(lambda s: _div(s.riichi_houjuu, s.riichi))
    chasing_riichi_rate = property(lambda s: _div(s.chasing_riichi, s.riichi))
    riichi_chased_rate = property(lambda s: _div(s.riichi_got_chased, s.riichi))
    avg_riichi_jun = property(lambda s: _div(s.riichi_jun, s.riichi))
    avg_riichi_point = property(lambda s: _div(s.riichi_point, s.riichi))
    agari_rate_as_oya = property(lambda s: _div(s.agari_as_oya, s.oya))
    agari_as_oya_rate = property(lambda s: _div(s.agari_as_oya, s.agari))
    houjuu_to_oya_rate = property(lambda s: _div(s.houjuu_to_oya, s.houjuu))
    avg_fuuro_num = property(lambda s: _div(s.fuuro_num, s.fuuro))
    agari_rate_after_fuuro = property(lambda s: _div(s.fuuro_agari, s.fuuro))
    houjuu_rate_after_fuuro = property(lambda s: _div(s.fuuro_houjuu, s.fuuro))
    avg_fuuro_point = property(lambda s: _div(s.fuuro_point, s.fuuro))
    yakuman_rate = property(lambda s: _div(s.yakuman, s.round))
    nagashi_mangan_rate = property(lambda s: _div(s.nagashi_mangan, s.round))

    # ---- stat.rs:128-256 (Display) / __repr__ (Debug)
    def __str__(self):
        def f6(x):
            return "NaN" if math.isnan(x) else ("inf" if x == math.inf else "-inf" if x == -math.inf else f"{x:.6f}")

        def f9(x):
            return "NaN" if math.isnan(x) else ("inf" if x == math.inf else "-inf" if x == -math.inf else f"{x:.9f}")

        s = self
        pts = [90, 45, 0, -135]
        return "\n".join([
            f"Games            {s.game}", f"Rounds           {s.round}", f"Rounds as dealer {s.oya}", "",
            f"1st (rate)       {s.rank_1} ({f6(s.rank_1_rate)})", f"2nd (rate)       {s.rank_2} ({f6(s.rank_2_rate)})",
            f"3rd (rate)       {s.rank_3} ({f6(s.rank_3_rate)})", f"4th (rate)       {s.rank_4} ({f6(s.rank_4_rate)})",
            f"Tobi(rate)       {s.tobi} ({f6(s.tobi_rate)})", f"Avg rank         {f6(s.avg_rank)}",
            f"Total rank pt    {s.total_pt(pts)}", f"Avg rank pt      {f6(s.avg_pt(pts))}",
            f"Total Δscore     {s.point}", f"Avg game Δscore  {f6(s.avg_point_per_game)}",
            f"Avg round Δscore {f6(s.avg_point_per_round)}", "",
            f"Win rate      {f6(s.agari_rate)}", f"Deal-in rate  {f6(s.houjuu_rate)}", f"Call rate     {f6(s.fuuro_rate)}",
            f"Riichi rate   {f6(s.riichi_rate)}", f"Ryukyoku rate {f6(s.ryukyoku_rate)}", "",
            f"Avg winning Δscore               {f6(s.avg_point_per_agari)}",
            f"Avg winning Δscore as dealer     {f6(s.avg_point_per_oya_agari)}",
            f"Avg winning Δscore as non-dealer {f6(s.avg_point_per_ko_agari)}",
            f"Avg riichi winning Δscore        {f6(s.avg_point_per_riichi_agari)}",
            f"Avg open winning Δscore          {f6(s.avg_point_per_fuuro_agari)}",
            f"Avg dama winning Δscore          {f6(s.avg_point_per_dama_agari)}",
            f"Avg ryukyoku Δscore              {f6(s.avg_point_per_ryukyoku)}", "",
            f"Avg winning turn        {f6(s.avg_agari_jun)}", f"Avg riichi winning turn {f6(s.avg_riichi_agari_jun)}",
            f"Avg open winning turn   {f6(s.avg_fuuro_agari_jun)}", f"Avg dama winning turn   {f6(s.avg_dama_agari_jun)}", "",
            f"Avg deal-in turn                 {f6(s.avg_houjuu_jun)}",
            f"Avg deal-in Δscore               {f6(s.avg_point_per_houjuu)}",
            f"Avg deal-in Δscore to dealer     {f6(s.avg_point_per_houjuu_to_oya)}",
            f"Avg deal-in Δscore to non-dealer {f6(s.avg_point_per_houjuu_to_ko)}", "",
            f"Chasing riichi rate       {f6(s.chasing_riichi_rate)}", f"Riichi chased rate        {f6(s.riichi_chased_rate)}",
            f"Winning rate after riichi {f6(s.agari_rate_after_riichi)}",
            f"Deal-in rate after riichi {f6(s.houjuu_rate_after_riichi)}", f"Avg riichi turn           {f6(s.avg_riichi_jun)}",
            f"Avg riichi Δscore         {f6(s.avg_riichi_point)}", "",
            f"Avg number of calls     {f6(s.avg_fuuro_num)}", f"Winning rate after call {f6(s.agari_rate_after_fuuro)}",
            f"Deal-in rate after call {f6(s.houjuu_rate_after_fuuro)}", f"Avg call Δscore         {f6(s.avg_fuuro_point)}", "",
            f"Dealer wins/all dealer rounds  {f6(s.agari_rate_as_oya)}", f"Dealer wins/all wins           {f6(s.agari_as_oya_rate)}",
            f"Deal-in to dealer/all deal-ins {f6(s.houjuu_to_oya_rate)}", "",
            f"Yakuman (rate)        {s.yakuman} ({f9(s.yakuman_rate)})",
            f"Nagashi mangan (rate) {s.nagashi_mangan} ({f9(s.nagashi_mangan_rate)})",
        ])

    def __repr__(self):
        return "Stat { " + ", ".join(f"{f}: {getattr(self, f)}" for f in _FIELDS) + " }"
