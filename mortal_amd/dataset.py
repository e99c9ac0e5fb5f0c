"""libriichi.dataset: `Grp` (dataset/grp.rs) and `GameplayLoader` / `Gameplay` (dataset/gameplay.rs) on the MI355X pool.

`Grp` is a host-side reduction of the event stream.  `GameplayLoader` replays whole batches of logs on the device:
one table per log, every seat's PlayerState advanced by the step kernel's event handlers (`mj_k_replay`), and every
decision of a wanted seat encoded by the arena's obs/mask kernels — obs, mask, label and bookkeeping are the
reference's (gameplay.rs:239-443), produced for all logs of a batch at once instead of one rayon task per player.

`oracle=True` adds the invisible obs of dataset/invisible.rs: the wall of every kyoku is either rebuilt on the device from
the game's seed (`trust_seed=True`, logs written by this engine / the reference arena) or reconstructed from the log with
the never-seen tiles filled in at random — like the reference, that filler is not reproducible.
"""
import gzip
import json

import numpy as np
import torch

from . import mjai_log
from .pool import ACTION_SPACE, OBS_ROWS, TablePool

GRP_SIZE = 7


def _read_gz(path):
    with gzip.open(path, "rt") as f:
        return f.read()


def _parse(raw_log):
    try:
        return [json.loads(l) for l in raw_log.splitlines() if l.strip()]
    except json.JSONDecodeError as ex:
        raise ValueError(f"failed to parse log: {ex}") from ex


class Grp:
    """dataset/grp.rs:20-165: per-kyoku [grand_kyoku, honba, kyotaku, scores/10000 x4] (f64), final ranks and scores."""

    def __init__(self, feature=None, rank_by_player=(0, 0, 0, 0), final_scores=(0, 0, 0, 0)):
        self.feature = np.zeros((0, GRP_SIZE), dtype=np.float64) if feature is None else feature
        self.rank_by_player = list(rank_by_player)
        self.final_scores = list(final_scores)

    @staticmethod
    def load_events(events):
        game_info = []
        rank_by_player = None
        final_deltas = [0, 0, 0, 0]
        final_scores = [0, 0, 0, 0]
        for ev in reversed(events):
            t = ev["type"]
            if t in ("hora", "ryukyoku"):
                if rank_by_player is None:
                    ds = ev.get("deltas")
                    if ds is None:
                        raise ValueError("invalid log: field `deltas` is required for Hora and Ryukyoku of AL")
                    final_deltas = [a + b for a, b in zip(final_deltas, ds)]
            elif t == "reach_accepted":
                if rank_by_player is None:
                    final_deltas[ev["actor"]] -= 1000
            elif t == "start_kyoku":
                if rank_by_player is None:
                    final_scores = [a + b for a, b in zip(ev["scores"], final_deltas)]
                    order = sorted(range(4), key=lambda i: -final_scores[i])  # Rankings::new (stable)
                    total = sum(final_scores)
                    if total < 100_000:  # assume the sum of scores to be 100k
                        final_scores[order[0]] += 100_000 - total
                    rank_by_player = [0] * 4
                    for r, pid in enumerate(order):
                        rank_by_player[pid] = r
                kyoku = ev["kyoku"]
                grand = {"E": kyoku - 1, "S": 3 + kyoku}.get(ev["bakaze"], 7 + kyoku)
                game_info.insert(0, [float(grand), float(ev["honba"]), float(ev["kyotaku"])]
                                 + [s / 10000.0 for s in ev["scores"]])
        if rank_by_player is None:
            raise ValueError("invalid log: no Hora or Ryukyoku after a StartKyoku")
        return Grp(np.array(game_info, dtype=np.float64).reshape(len(game_info), GRP_SIZE), rank_by_player, final_scores)

    @staticmethod
    def load_log(raw_log):
        return Grp.load_events(_parse(raw_log))

    @staticmethod
    def load_gz_log_files(gzip_filenames):
        out = []
        for f in gzip_filenames:
            try:
                out.append(Grp.load_log(_read_gz(f)))
            except Exception as ex:
                raise RuntimeError(f"error when reading {f}: {ex}") from ex
        return out

    def __len__(self):
        return self.feature.shape[0]

    def take_feature(self):
        f, self.feature = self.feature, np.zeros((0, GRP_SIZE), dtype=np.float64)
        return f

    def take_rank_by_player(self):
        return list(self.rank_by_player)

    def take_final_scores(self):
        return list(self.final_scores)


class Gameplay:
    """One player's samples of one game (gameplay.rs:46-64).  The `take_*` methods hand the per-move lists over like
    the reference's; the same data stays available as stacked device tensors (`obs_dev`, `masks_dev`) for consumers
    that batch on the GPU anyway (mortal/dataloader.py stacks them right after)."""

    def __init__(self, player_id, player_name, grp):
        self.player_id = player_id
        self.player_name = player_name
        self.grp = grp
        self.obs_dev = None
        self.masks_dev = None
        self.invisible_obs = []
        self.actions = []
        self.at_kyoku = []
        self.dones = []
        self.apply_gamma = []
        self.at_turns = []
        self.shantens = []

    def _take_list(self, name):
        v = getattr(self, name)
        setattr(self, name, [])
        return v

    def take_obs(self):
        t, self.obs_dev = self.obs_dev, None
        return [] if t is None else list(t.cpu().numpy())

    def take_masks(self):
        t, self.masks_dev = self.masks_dev, None
        return [] if t is None else list(t.cpu().numpy())

    def take_invisible_obs(self):
        return self._take_list("invisible_obs")

    def take_actions(self):
        return self._take_list("actions")

    def take_at_kyoku(self):
        return self._take_list("at_kyoku")

    def take_dones(self):
        return self._take_list("dones")

    def take_apply_gamma(self):
        return self._take_list("apply_gamma")

    def take_at_turns(self):
        return self._take_list("at_turns")

    def take_shantens(self):
        return self._take_list("shantens")

    def take_grp(self):
        g, self.grp = self.grp, Grp()
        return g

    def take_player_id(self):
        return self.player_id


class GameplayLoader:
    """dataset/gameplay.rs:21-165."""

    pool_cls = TablePool  # the test suite substitutes the host emulation of the same kernels (tests/host/emu_pool.py)

    def __init__(self, version, *, oracle=True, player_names=None, excludes=None, trust_seed=False,
                 always_include_kan_select=True, augmented=False, device="cuda:0", deal_algo=None):
        if version not in OBS_ROWS:
            raise ValueError(f"unsupported obs version {version}")
        self.version = version
        self.oracle = bool(oracle)
        self.player_names = list(player_names or [])
        self.excludes = list(excludes or [])
        self.trust_seed = bool(trust_seed)
        self.always_include_kan_select = bool(always_include_kan_select)
        self.augmented = bool(augmented)
        self.device = device
        # trust_seed: wall shuffle tried first when rebuilding a kyoku from its seed (None = pool.default_deal_algo());
        # the replay kernel falls back to the other rand generation before reporting MJ_ERR_WALL
        self.deal_algo = deal_algo
        # oracle + trust_seed + augmented: the reference augments the events but deals the invisible wall from the seed as it
        # was (gameplay.rs:126-164, invisible.rs:36-71); the replay kernel does the same (LG_SK_AUG_BIT)

    def __repr__(self):
        return (f"GameplayLoader {{ version: {self.version}, oracle: {self.oracle}, player_names: {self.player_names}, "
                f"excludes: {self.excludes}, trust_seed: {self.trust_seed}, "
                f"always_include_kan_select: {self.always_include_kan_select}, augmented: {self.augmented} }}")

    # ---- the reference's entry points
    def load_log(self, raw_log):
        return self.load_logs([raw_log])[0]

    def load_gz_log_files(self, gzip_filenames):
        raws = []
        for f in gzip_filenames:
            try:
                raws.append(_read_gz(f))
            except Exception as ex:
                raise RuntimeError(f"error when reading {f}: {ex}") from ex
        return self.load_logs(raws)

    def _wanted(self, names):
        ps, ex = set(self.player_names), set(self.excludes)
        out = []
        for i, name in enumerate(names):
            if ps:
                keep = name in ps
            elif ex:
                keep = name not in ex
            else:
                keep = True
            if keep:
                out.append(i)
        return out

    @staticmethod
    def _walls_from_events(events, augmented, rng):
        """Invisible::new without a seed (invisible.rs:24-149): per kyoku the 136-tile wall in the pool's layout —
        haipai 0..51, rinshan 52..55 (popped from the back), dora indicators 56..60 (back first), ura 61..65, yama 66..135
        (popped from the back) — known tiles from the log, the rest drawn at random from the unseen tiles."""
        tid = (lambda n: mjai_log.augment_tile_id(mjai_log.TILE_ID[n])) if augmented else (lambda n: mjai_log.TILE_ID[n])
        walls = []
        cur = None

        def fresh():
            unknown = [4] * 37
            for t in (4, 13, 22):
                unknown[t] = 3
            unknown[34] = unknown[35] = unknown[36] = 1
            return dict(hai=[], yama=[], rinshan=[], dora=[], ura=[], unknown=unknown, from_rinshan=False, ura_done=False)

        for ev in events:
            t = ev["type"]
            if t == "start_kyoku":
                cur = fresh()
                cur["dora"].append(tid(ev["dora_marker"]))
                cur["unknown"][cur["dora"][0]] -= 1
                for hand in ev["tehais"]:
                    for x in hand:
                        cur["hai"].append(tid(x))
                        cur["unknown"][tid(x)] -= 1
            elif cur is None:
                continue
            elif t == "tsumo":
                p = tid(ev["pai"])
                if cur["from_rinshan"]:
                    cur["rinshan"].append(p)
                    cur["from_rinshan"] = False
                else:
                    cur["yama"].append(p)
                cur["unknown"][p] -= 1
            elif t in ("ankan", "kakan", "daiminkan"):
                cur["from_rinshan"] = True
            elif t == "dora":
                cur["dora"].append(tid(ev["dora_marker"]))
                cur["unknown"][cur["dora"][-1]] -= 1
            elif t == "hora" and ev.get("ura_markers") is not None and not cur["ura_done"]:
                for x in ev["ura_markers"]:
                    cur["ura"].append(tid(x))
                    cur["unknown"][tid(x)] -= 1
                cur["ura_done"] = True
            elif t == "end_kyoku":
                if min(cur["unknown"]) < 0:
                    raise ValueError("invalid log: more than four copies of a tile")
                filler = [k for k, c in enumerate(cur["unknown"]) for _ in range(c)]
                rng.shuffle(filler)
                for key, size in (("yama", 70), ("rinshan", 4), ("dora", 5), ("ura", 5)):
                    while len(cur[key]) < size:
                        cur[key].append(filler.pop())
                assert not filler
                wall = [0] * 136
                wall[:52] = cur["hai"]
                for k in range(4):
                    wall[52 + 3 - k] = cur["rinshan"][k]
                for k in range(5):
                    wall[56 + 4 - k] = cur["dora"][k]
                    wall[61 + k] = cur["ura"][k]
                for k in range(70):
                    wall[66 + 69 - k] = cur["yama"][k]
                walls.append(wall)
                cur = None
        return walls

    def load_logs(self, raw_logs):
        """Batch entry point: list of raw log texts -> list (per log) of lists of Gameplay (one per wanted player)."""
        games = []
        for raw in raw_logs:
            events = _parse(raw)
            if not events or events[0].get("type") != "start_game":
                raise ValueError("empty or invalid game log")
            names = events[0].get("names", ["", "", "", ""])
            wanted = self._wanted(names)
            games.append(dict(events=events, names=names, wanted=wanted, grp=Grp.load_events(events)))
        n = len(games)
        if n == 0:
            return []
        nonces = keys = None
        if not self.oracle:
            scripts = [mjai_log.encode_events(g["events"], augmented=self.augmented) for g in games]
        else:
            rng = np.random.default_rng()
            seeds = [g["events"][0].get("seed") if self.trust_seed else None for g in games]
            nonces = np.array([s[0] if s else 0 for s in seeds], dtype=np.uint64)
            keys = np.array([s[1] if s else 0 for s in seeds], dtype=np.uint64)
            scripts = []
            for g, seed in zip(games, seeds):
                if seed:  # the game was emulated by this engine: use the seed directly (invisible.rs:36-71)
                    scripts.append(mjai_log.encode_events(g["events"], augmented=self.augmented, deal_from_seed=True))
                else:
                    walls = self._walls_from_events(g["events"], self.augmented, rng)
                    scripts.append(mjai_log.encode_events(g["events"], augmented=self.augmented, walls=walls))
        tracked = [sum(1 << p for p in g["wanted"]) for g in games]
        total_events = sum(len(g["events"]) for g in games)
        pool = self.pool_cls(n, version=self.version, device=self.device, max_rows=8 * n + 64, deal_algo=self.deal_algo)
        try:
            pool.replay_load(scripts, tracked, self.always_include_kan_select, nonces, keys)
            obs_parts, mask_parts, meta_parts, inv_parts = [], [], [], []
            for _ in range(total_events + 8):
                k = pool.replay_step()
                if k == 0:
                    if pool.counters()["games"] >= n:
                        break
                    continue
                obs, masks = pool.encode(0)
                obs_parts.append(obs)
                mask_parts.append(masks)
                meta_parts.append(pool.replay_meta())
                if self.oracle:
                    inv_parts.append(pool.encode_oracle(0))
            else:
                raise RuntimeError("log replay did not terminate")
            code, tbl = pool.first_error()
            if code:
                raise ValueError(f"log {tbl}: the event stream is not a legal game (error code {code})")
            if pool.counters()["sp_overflow"]:
                raise RuntimeError("obs v4: a sample's single-player state graph exceeded the device scratch capacity")
        finally:
            pool.close()
        C = OBS_ROWS[self.version]
        if obs_parts:
            obs = torch.cat(obs_parts)
            masks = torch.cat(mask_parts)
            meta = torch.cat(meta_parts).cpu().numpy()
        else:
            obs = torch.empty((0, C, 34), dtype=torch.float32, device=self.device)
            masks = torch.empty((0, ACTION_SPACE), dtype=torch.bool, device=self.device)
            meta = np.zeros((0, 8), dtype=np.int32)
        # order the samples of every (log, seat) like the reference: by event, the kan-select entry after its main entry
        order = np.lexsort((meta[:, 6], meta[:, 7], meta[:, 2], meta[:, 1])) if len(meta) else np.zeros(0, dtype=np.int64)
        meta = meta[order]
        idx_dev = torch.as_tensor(order, device=obs.device, dtype=torch.long)
        obs, masks = obs[idx_dev], masks[idx_dev]
        inv = torch.cat(inv_parts)[idx_dev].cpu().numpy() if (self.oracle and inv_parts) else None
        out = []
        pos = 0
        for t, g in enumerate(games):
            per_log = []
            for p in g["wanted"]:
                lo = pos
                while pos < len(meta) and meta[pos, 1] == t and meta[pos, 2] == p:
                    pos += 1
                m = meta[lo:pos]
                gp = Gameplay(p, g["names"][p], Grp(g["grp"].feature.copy(), g["grp"].rank_by_player, g["grp"].final_scores))
                gp.obs_dev = obs[lo:pos]
                gp.masks_dev = masks[lo:pos]
                if inv is not None:
                    gp.invisible_obs = list(inv[lo:pos])
                gp.actions = [int(x) for x in m[:, 0]]
                gp.at_kyoku = [int(x) for x in m[:, 3]]
                gp.at_turns = [int(x) for x in m[:, 4]]
                gp.shantens = [int(x) for x in m[:, 5]]
                gp.apply_gamma = [bool(x <= 37) for x in m[:, 0]]  # only discard and kan will discount (gameplay.rs:423)
                ak = gp.at_kyoku
                gp.dones = [ak[i + 1] > ak[i] for i in range(len(ak) - 1)] + [True]  # gameplay.rs:264-265
                per_log.append(gp)
            out.append(per_log)
        assert pos == len(meta)
        return out
