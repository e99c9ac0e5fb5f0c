"""Device event log -> mjai JSON lines (+ .json.gz dump).

The step kernel appends one u64 header word (+ payload words) per event to a per-table log
(mortal_amd/csrc/mj_state.h LG_*).  This module turns those words into the reference's serialisation:
`Event` of mjai/event.rs:14-121 (serde: tag "type", snake_case, declaration field order, compact separators) and
`GameResult::dump_json_log` (arena/result.rs:32-51): a start_game line with names and seed, every event of every kyoku,
an end_game line.  File naming and gzip follow arena/one_vs_three.rs:195-225.
"""
import gzip
import json
import os

TILE_NAMES = ([f"{n}{s}" for s in "mps" for n in range(1, 10)] + ["E", "S", "W", "N", "P", "F", "C", "5mr", "5pr", "5sr", "?"])

(LG_START_KYOKU, LG_TSUMO, LG_DAHAI, LG_CHI, LG_PON, LG_DAIMINKAN, LG_KAKAN, LG_ANKAN, LG_DORA, LG_REACH,
 LG_REACH_ACCEPTED, LG_HORA, LG_RYUKYOKU, LG_END_KYOKU) = range(1, 15)
_NURA_SHIFT, _HONBA_SHIFT, _KYOTAKU_SHIFT = 39, 44, 52
_TAG_BIT = 43


def _i32x4(w0, w1):
    out = []
    for w in (w0, w1):
        for sh in (0, 32):
            v = (w >> sh) & 0xFFFFFFFF
            out.append(v - (1 << 32) if v & 0x80000000 else v)
    return out


def decode_events(words, tags=None):
    """u64 words of one table -> list of mjai event dicts (keys in the reference's serialisation order).

    Arena logs mark an agent's reactions with a tag word (LG_TAG_BIT); pass a list as `tags` to receive, per event,
    None or dict(cycle, row, kan_row, shanten, at_furiten) — the link to the decision's batch row for the `meta` object."""
    evs = []
    i, n = 0, len(words)
    tn = TILE_NAMES
    while i < n:
        w = int(words[i])
        i += 1
        t = w & 15
        actor, target = (w >> 4) & 3, (w >> 6) & 3
        pai = (w >> 8) & 63
        c = [(w >> (14 + 6 * k)) & 63 for k in range(4)]
        tag = None
        if t != LG_START_KYOKU and (w >> _TAG_BIT) & 1:
            tw = int(words[i])
            i += 1
            kr = (tw >> 38) & 0x3FFFF
            tag = dict(cycle=tw & 0xFFFFF, row=(tw >> 20) & 0x3FFFF, kan_row=kr - 1 if kr else None,
                       shanten=((tw >> 56) & 15) - 1, at_furiten=bool((tw >> 60) & 1))
        if tags is not None:
            tags.append(tag)
        if t == LG_START_KYOKU:
            kyoku = c[0]
            scores = _i32x4(int(words[i]), int(words[i + 1]))
            tiles = []
            for k in range(7):
                v = int(words[i + 2 + k])
                tiles += [(v >> (8 * b)) & 0xFF for b in range(8)]
            i += 9 + (17 if (w >> 63) & 1 else 0)
            evs.append({"type": "start_kyoku", "bakaze": tn[27 + kyoku // 4], "dora_marker": tn[pai], "kyoku": kyoku % 4 + 1,
                        "honba": (w >> _HONBA_SHIFT) & 0xFF, "kyotaku": (w >> _KYOTAKU_SHIFT) & 0xFF, "oya": kyoku % 4,
                        "scores": scores, "tehais": [[tn[x] for x in tiles[s * 13:(s + 1) * 13]] for s in range(4)]})
        elif t == LG_TSUMO:
            evs.append({"type": "tsumo", "actor": actor, "pai": tn[pai]})
        elif t == LG_DAHAI:
            evs.append({"type": "dahai", "actor": actor, "pai": tn[pai], "tsumogiri": bool((w >> 38) & 1)})
        elif t in (LG_CHI, LG_PON):
            evs.append({"type": "chi" if t == LG_CHI else "pon", "actor": actor, "target": target, "pai": tn[pai],
                        "consumed": [tn[c[0]], tn[c[1]]]})
        elif t == LG_DAIMINKAN:
            evs.append({"type": "daiminkan", "actor": actor, "target": target, "pai": tn[pai],
                        "consumed": [tn[c[0]], tn[c[1]], tn[c[2]]]})
        elif t == LG_KAKAN:
            evs.append({"type": "kakan", "actor": actor, "pai": tn[pai], "consumed": [tn[c[0]], tn[c[1]], tn[c[2]]]})
        elif t == LG_ANKAN:
            evs.append({"type": "ankan", "actor": actor, "consumed": [tn[x] for x in c]})
        elif t == LG_DORA:
            evs.append({"type": "dora", "dora_marker": tn[pai]})
        elif t == LG_REACH:
            evs.append({"type": "reach", "actor": actor})
        elif t == LG_REACH_ACCEPTED:
            evs.append({"type": "reach_accepted", "actor": actor})
        elif t == LG_HORA:
            deltas = _i32x4(int(words[i]), int(words[i + 1]))
            u = int(words[i + 2])
            i += 3
            n_ura = (w >> _NURA_SHIFT) & 7
            evs.append({"type": "hora", "actor": actor, "target": target, "deltas": deltas,
                        "ura_markers": [tn[(u >> (6 * k)) & 63] for k in range(n_ura)]})
        elif t == LG_RYUKYOKU:
            deltas = _i32x4(int(words[i]), int(words[i + 1]))
            i += 2
            evs.append({"type": "ryukyoku", "deltas": deltas})
        elif t == LG_END_KYOKU:
            evs.append({"type": "end_kyoku"})
        else:
            raise ValueError(f"corrupt event log word {w:#x} at {i - 1}")
    return evs


TILE_ID = {n: i for i, n in enumerate(TILE_NAMES)}


def augment_tile_id(t):
    """Tile::augment (tile.rs:154-167): swap manzu and pinzu, keep the red flag."""
    if t >= 37:
        return t
    aka = t >= 34
    d = (4, 13, 22)[t - 34] if aka else t
    d = d + 9 if d < 9 else d - 9 if d < 18 else d
    return {4: 34, 13: 35, 22: 36}[d] if aka else d


def encode_events(events, augmented=False, walls=None, deal_from_seed=False):
    """mjai event dicts (start_game / end_game skipped) -> uint64 words in the LG_* format (inverse of decode_events).

    Replay scripts only: `walls` = one 136-tile id sequence per kyoku (appended to its start_kyoku, LG_SK_WALL_BIT), or
    `deal_from_seed` (LG_SK_DEAL_BIT: the device rebuilds the wall from the table's seed)."""
    import numpy as np

    tid = (lambda name: augment_tile_id(TILE_ID[name])) if augmented else (lambda name: TILE_ID[name])

    def word(t, actor=0, target=0, pai=0, c=(), tsumogiri=0):
        w = t | (actor << 4) | (target << 6) | ((pai & 63) << 8) | (int(bool(tsumogiri)) << 38)
        for k, x in enumerate(c):
            w |= (x & 63) << (14 + 6 * k)
        return w

    def i32x2(a, b):
        return (a & 0xFFFFFFFF) | ((b & 0xFFFFFFFF) << 32)

    out = []
    n_kyoku = 0
    for e in events:
        t = e["type"]
        if t in ("start_game", "end_game", "none"):
            continue
        # serde's bounded integers (mjai/event.rs:20-120, `BoundedU8<0, 3>` actors, kyoku 1..=4): reject, never wrap
        for k in ("actor", "target", "oya"):
            if k in e and not 0 <= int(e[k]) <= 3:
                raise ValueError(f"{t}: {k} {e[k]} out of range 0..3")
        if t == "start_kyoku" and not (1 <= int(e["kyoku"]) <= 4 and 0 <= int(e["honba"]) <= 255 and 0 <= int(e["kyotaku"]) <= 255
                                       and e["bakaze"] in ("E", "S", "W", "N")):
            raise ValueError("start_kyoku: kyoku / honba / kyotaku / bakaze out of range")
        if t == "start_kyoku":
            kyoku = (TILE_ID[e["bakaze"]] - 27) * 4 + e["kyoku"] - 1
            out.append(word(LG_START_KYOKU, pai=tid(e["dora_marker"]), c=(kyoku,)) | (e["honba"] << _HONBA_SHIFT)
                       | (e["kyotaku"] << _KYOTAKU_SHIFT))
            sc = e["scores"]
            out += [i32x2(sc[0], sc[1]), i32x2(sc[2], sc[3])]
            tiles = [tid(x) for hand in e["tehais"] for x in hand]
            if len(tiles) != 52:
                raise ValueError("start_kyoku needs 4 x 13 tiles")
            for k in range(7):
                out.append(sum((tiles[k * 8 + b] if k * 8 + b < 52 else 0) << (8 * b) for b in range(8)))
            if deal_from_seed:
                out[-10] |= (1 << 62) | ((1 << 61) if augmented else 0)  # LG_SK_DEAL_BIT, LG_SK_AUG_BIT
            elif walls is not None:
                wall = [int(x) for x in walls[n_kyoku]]
                if len(wall) != 136 or wall[:52] != tiles:
                    raise ValueError("wall does not start with the logged haipai")
                out[-10] |= 1 << 63
                for k in range(17):
                    out.append(sum(wall[k * 8 + b] << (8 * b) for b in range(8)))
            n_kyoku += 1
        elif t == "tsumo":
            out.append(word(LG_TSUMO, e["actor"], pai=tid(e["pai"])))
        elif t == "dahai":
            out.append(word(LG_DAHAI, e["actor"], pai=tid(e["pai"]), tsumogiri=e["tsumogiri"]))
        elif t in ("chi", "pon", "daiminkan"):
            code = {"chi": LG_CHI, "pon": LG_PON, "daiminkan": LG_DAIMINKAN}[t]
            out.append(word(code, e["actor"], e["target"], tid(e["pai"]), [tid(x) for x in e["consumed"]]))
        elif t == "kakan":
            out.append(word(LG_KAKAN, e["actor"], pai=tid(e["pai"]), c=[tid(x) for x in e["consumed"]]))
        elif t == "ankan":
            out.append(word(LG_ANKAN, e["actor"], c=[tid(x) for x in e["consumed"]]))
        elif t == "dora":
            out.append(word(LG_DORA, pai=tid(e["dora_marker"])))
        elif t == "reach":
            out.append(word(LG_REACH, e["actor"]))
        elif t == "reach_accepted":
            out.append(word(LG_REACH_ACCEPTED, e["actor"]))
        elif t == "hora":
            ura = [tid(x) for x in (e.get("ura_markers") or [])]
            d = e.get("deltas") or [0, 0, 0, 0]
            out.append(word(LG_HORA, e["actor"], e["target"]) | (len(ura) << _NURA_SHIFT))
            out += [i32x2(d[0], d[1]), i32x2(d[2], d[3]), sum(u << (6 * k) for k, u in enumerate(ura))]
        elif t == "ryukyoku":
            d = e.get("deltas") or [0, 0, 0, 0]
            out += [word(LG_RYUKYOKU), i32x2(d[0], d[1]), i32x2(d[2], d[3])]
        elif t == "end_kyoku":
            out.append(word(LG_END_KYOKU))
        else:
            raise ValueError(f"unknown event type {t!r}")
    return np.array(out, dtype=np.uint64)


def _dumps(ev):
    return json.dumps(ev, separators=(",", ":"), ensure_ascii=False)


def dump_json_log(names, seed, events):
    """arena/result.rs:32-51."""
    lines = [_dumps({"type": "start_game", "names": list(names), "seed": [int(seed[0]), int(seed[1])]})]
    lines += [_dumps(e) for e in events]
    lines.append(_dumps({"type": "end_game"}))
    return "\n".join(lines) + "\n"


def write_game_log_as(path, names, seed, events):
    """One `{seed}_{key}_{split}.json.gz` per game (one_vs_three.rs:213-222); gzip level 9 = Compression::best()."""
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
        f.write(dump_json_log(names, seed, events).encode())
    return path


def read_game_log(path):
    """-> list of event dicts of one `.json` / `.json.gz` log file."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return [json.loads(l) for l in f if l.strip()]
