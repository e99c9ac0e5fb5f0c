"""Test infrastructure: the reference's dataset loader (dataset/gameplay.rs:239-443 Gameplay::load_events_by_player),
restated on top of the oracle's PlayerState.  Used only to check mortal_amd.dataset.GameplayLoader."""
import numpy as np

_DEAKA = {"5mr": "5m", "5pr": "5p", "5sr": "5s"}


def _tile_id(O, name):
    return O.TILE_ID[_DEAKA.get(name, name)]


def load_events_by_player(O, events, player_id, version, always_include_kan_select=True):
    st = O.PlayerState(player_id)
    kyoku_idx = 0
    out = dict(obs=[], masks=[], actions=[], at_kyoku=[], apply_gamma=[], at_turns=[], shantens=[], player_name="")

    def add_entry(at_kan, label):
        obs, mask = st.encode_obs(version, at_kan)
        sn = st.snapshot()
        out["obs"].append(obs)
        out["masks"].append(mask)
        out["actions"].append(label)
        out["at_kyoku"].append(kyoku_idx)
        out["apply_gamma"].append(label <= 37)
        out["at_turns"].append(sn["at_turn"])
        out["shantens"].append(sn["shanten"])

    for i in range(len(events) - 3):  # events.windows(4)
        wnd = events[i:i + 4]
        cur = wnd[0]
        nxt = wnd[2] if wnd[1]["type"] in ("reach_accepted", "dora") else wnd[1]
        if cur["type"] == "start_game":
            out["player_name"] = cur["names"][player_id]
            continue  # PlayerState::update ignores it
        if cur["type"] == "end_kyoku":
            kyoku_idx += 1
        cans = st.update(cur)
        can_chi = cans["can_chi_low"] or cans["can_chi_mid"] or cans["can_chi_high"]
        can_act = (cans["can_discard"] or can_chi or cans["can_pon"] or cans["can_daiminkan"] or cans["can_kakan"]
                   or cans["can_ankan"] or cans["can_riichi"] or cans["can_tsumo_agari"] or cans["can_ron_agari"]
                   or cans["can_ryukyoku"])
        if not can_act:
            continue
        kan_select = None
        label = None
        t = nxt["type"]
        if t == "dahai":
            label = O.TILE_ID[nxt["pai"]]
        elif t == "reach":
            label = 37
        elif t == "chi" and nxt["actor"] == player_id:
            a, b = sorted(_tile_id(O, x) for x in nxt["consumed"])
            p = _tile_id(O, nxt["pai"])
            label = 38 if p < a else 39 if p < b else 40
        elif t == "pon" and nxt["actor"] == player_id:
            label = 41
        elif t == "daiminkan" and nxt["actor"] == player_id:
            if always_include_kan_select:
                kan_select = _tile_id(O, nxt["pai"])
            label = 42
        elif t == "kakan":
            if always_include_kan_select or st.snapshot()["n_kakan_cand"] > 1:
                kan_select = _tile_id(O, nxt["pai"])
            label = 42
        elif t == "ankan":
            if always_include_kan_select or st.snapshot()["n_ankan_cand"] > 1:
                kan_select = _tile_id(O, nxt["consumed"][0])
            label = 42
        elif t == "ryukyoku" and cans["can_ryukyoku"]:
            label = 44
        else:
            has_any_ron = wnd[1]["type"] == "hora"
            if has_any_ron:
                for ev in wnd[1:]:
                    if ev["type"] == "end_kyoku":
                        break
                    if ev["type"] == "hora" and ev["actor"] == player_id:
                        label = 43
                        break
            if label is None:
                if (can_chi and t == "tsumo") or ((cans["can_pon"] or cans["can_daiminkan"] or cans["can_ron_agari"])
                                                  and not has_any_ron):
                    label = 45
        if label is not None:
            add_entry(False, label)
            if kan_select is not None:
                add_entry(True, kan_select)
    ak = out["at_kyoku"]
    out["dones"] = [ak[k + 1] > ak[k] for k in range(len(ak) - 1)] + [True]
    return out


# ---------------------------------------------------------------- dataset/invisible.rs (trust_seed path) restated
def invisibles_from_seed(O, events):
    """Invisible::new with trust_seed=True (invisible.rs:36-71): per kyoku (yama, rinshan, dora, ura), early -> late."""
    nonce, key = events[0]["seed"]
    out = []
    for ev in events:
        if ev["type"] != "start_kyoku":
            continue
        kyoku = 4 * (O.TILE_ID[ev["bakaze"]] - 27) + ev["kyoku"] - 1
        seq = [int(x) for x in O.deal(nonce, key, kyoku, ev["honba"], algo=0)]
        # board.rs:111-122: haipai 0..52, rinshan 52..56, dora 56..61, ura 61..66, yama 66..136; rinshan / dora / yama pop
        out.append(dict(yama=seq[66:136][::-1], rinshan=seq[52:56][::-1], dora=seq[56:61][::-1], ura=seq[61:66]))
    return out


def invisible_encode(O, inv, opponents, yama_idx, rinshan_idx, version):
    """Invisible::encode (invisible.rs:153-243)."""
    rows = 211 if version == 1 else 217
    arr = np.zeros((rows, 34), dtype=np.float32)
    idx = 0
    for st in opponents:
        sn = st.snapshot()
        for t in range(34):
            arr[idx:idx + int(sn["tehai"][t]), t] = 1.0
        idx += 4
        for i in range(3):
            if sn["akas_in_hand"][i]:
                arr[idx + i, :] = 1.0
        idx += 3
        n = sn["shanten"]
        if version == 1:
            arr[idx:idx + n, :] = 1.0
            idx += 6
        else:
            arr[idx + n, :] = 1.0
            idx += 7
            arr[idx, :] = np.float32(n) / np.float32(6.0)
            idx += 1
        arr[idx, sn["waits"]] = 1.0
        idx += 1
        if sn["at_furiten"]:
            arr[idx, :] = 1.0
        idx += 1

    def encode_tile(r, tile):
        arr[r, {34: 4, 35: 13, 36: 22}.get(tile, tile)] = 1.0
        if tile >= 34:
            arr[r + 1, :] = 1.0

    for tile in inv["yama"][yama_idx:]:
        encode_tile(idx, tile)
        idx += 2
    idx += (yama_idx - 1) * 2
    for tile in inv["rinshan"][rinshan_idx:]:
        encode_tile(idx, tile)
        idx += 2
    idx += rinshan_idx * 2
    for tile in inv["dora"]:
        encode_tile(idx, tile)
        idx += 2
    for tile in inv["ura"]:
        encode_tile(idx, tile)
        idx += 2
    assert idx == rows
    return arr


def load_invisible_by_player(O, events, player_id, version, always_include_kan_select=True):
    """The invisible_obs list of Gameplay::load_events_by_player with oracle=True, trust_seed=True."""
    invs = invisibles_from_seed(O, events)
    opp = [O.PlayerState((player_id + i + 1) % 4) for i in range(3)]
    ref = load_events_by_player(O, events, player_id, version, always_include_kan_select)
    out = []
    kyoku_idx = yama_idx = rinshan_idx = 0
    from_rinshan = False
    # walk the events, mirroring the cursor bookkeeping of gameplay.rs:281-305
    entries_at = {}
    for i in range(len(events) - 3):
        cur = events[i]
        if cur["type"] == "start_game":
            continue
        if cur["type"] == "end_kyoku":
            kyoku_idx += 1
            from_rinshan = False
            yama_idx = rinshan_idx = 0
        elif cur["type"] == "tsumo":
            if from_rinshan:
                rinshan_idx += 1
                from_rinshan = False
            else:
                yama_idx += 1
        elif cur["type"] in ("ankan", "kakan", "daiminkan"):
            from_rinshan = True
        for s in opp:
            s.update(cur)
        entries_at[i] = (kyoku_idx, yama_idx, rinshan_idx, [s.clone() for s in opp])
    # which events produced entries: replay the label logic through load_events_by_player's own bookkeeping
    marks = entry_event_indices(O, events, player_id, always_include_kan_select)
    for i, n_rows in marks:
        k, y, r, opps = entries_at[i]
        enc = invisible_encode(O, invs[k], opps, y, r, version)
        out += [enc] * n_rows
    assert len(out) == len(ref["actions"])
    return out


def entry_event_indices(O, events, player_id, always_include_kan_select=True):
    """[(event index, number of entries)] for every event after which load_events_by_player adds entries."""
    st = O.PlayerState(player_id)
    marks = []
    for i in range(len(events) - 3):
        cur = events[i]
        if cur["type"] == "start_game":
            continue
        cans = st.update(cur)
        before = _count_entries(O, st, cans, events[i:i + 4], player_id, always_include_kan_select)
        if before:
            marks.append((i, before))
    return marks


def _count_entries(O, st, cans, wnd, player_id, always_kan):
    """Number of entries (0, 1 or 2) gameplay.rs:296-409 adds for this window."""
    can_chi = cans["can_chi_low"] or cans["can_chi_mid"] or cans["can_chi_high"]
    if not (cans["can_discard"] or can_chi or cans["can_pon"] or cans["can_daiminkan"] or cans["can_kakan"] or cans["can_ankan"]
            or cans["can_riichi"] or cans["can_tsumo_agari"] or cans["can_ron_agari"] or cans["can_ryukyoku"]):
        return 0
    nxt = wnd[2] if wnd[1]["type"] in ("reach_accepted", "dora") else wnd[1]
    t = nxt["type"]
    if t in ("dahai", "reach"):
        return 1
    if t in ("chi", "pon") and nxt["actor"] == player_id:
        return 1
    if t == "daiminkan" and nxt["actor"] == player_id:
        return 2 if always_kan else 1
    if t == "kakan":
        return 2 if (always_kan or st.snapshot()["n_kakan_cand"] > 1) else 1
    if t == "ankan":
        return 2 if (always_kan or st.snapshot()["n_ankan_cand"] > 1) else 1
    if t == "ryukyoku" and cans["can_ryukyoku"]:
        return 1
    has_any_ron = wnd[1]["type"] == "hora"
    if has_any_ron:
        for ev in wnd[1:]:
            if ev["type"] == "end_kyoku":
                break
            if ev["type"] == "hora" and ev["actor"] == player_id:
                return 1
    if (can_chi and t == "tsumo") or ((cans["can_pon"] or cans["can_daiminkan"] or cans["can_ron_agari"]) and not has_any_ron):
        return 1
    return 0
