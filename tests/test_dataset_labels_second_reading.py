"""A second, log-only reading of the dataset loader's labels (dataset/gameplay.rs:296-409).

tests/dataset_ref.py restates `Gameplay::load_events_by_player` on top of the oracle's PlayerState, and the device loader
(mortal_amd/csrc/mj_replay.hip) is compared with it sample by sample (tests/test_dataset.py, emulator twin).  Both were written by
reading the same Rust.  This test derives what it can from the mjai LOG ALONE, in a different shape — no PlayerState, no
window-of-four scan: the player's own actions in log order, mapped to action ids from the event, are exactly the non-pass labels of
the loader, in order (a kan followed by its kan-select sample when `always_include_kan_select`); the kyoku index of a sample is the
number of `end_kyoku` events before the action; `apply_gamma` is `label <= 37`; `dones` marks the last sample of every kyoku; every
pass sample (45) answers another seat's discard or kakan.  A misreading of the label table (chi low / mid / high by the called tile's
position, riichi = 37, kan = 42 + tile, hora = 43, kyuushu = 44), of the reach_accepted / dora skip, or of the kyoku bookkeeping
would have to be made twice.  CPU only."""
import json

import pytest

import dataset_ref
import parity_util
import test_dataset as TD

_DEAKA = {"5mr": "5m", "5pr": "5p", "5sr": "5s"}


def _own_actions(O, events, pid, always_kan):
    """(label, kyoku index, is_kan_select) of every action the log attributes to `pid`, in order."""
    tid = lambda name: O.TILE_ID[_DEAKA.get(name, name)]
    out, kyoku = [], 0
    n = len(events)
    for i, ev in enumerate(events):
        t = ev["type"]
        if t == "end_kyoku":
            kyoku += 1
            continue
        if ev.get("actor") != pid and t != "ryukyoku":
            continue
        if t == "dahai":
            out.append((O.TILE_ID[ev["pai"]], kyoku, False))  # red fives keep their own id (34..36)
        elif t == "reach":
            out.append((37, kyoku, False))
        elif t == "chi":
            lo, hi = sorted(tid(x) for x in ev["consumed"])
            p = tid(ev["pai"])
            out.append((38 if p < lo else 39 if p < hi else 40, kyoku, False))
        elif t == "pon":
            out.append((41, kyoku, False))
        elif t in ("daiminkan", "kakan", "ankan"):
            out.append((42, kyoku, False))
            if always_kan:
                out.append((tid(ev["pai"] if t != "ankan" else ev["consumed"][0]), kyoku, True))
        elif t == "hora":
            out.append((43, kyoku, False))
        elif t == "ryukyoku":
            # kyuushu kyuuhai is the only ryukyoku a player chooses: it follows that player's own draw directly
            prev = events[i - 1]
            if prev["type"] == "tsumo" and prev["actor"] == pid and "tenpais" not in ev and ev.get("reason", "kyuushu") in ("kyuushu", "kyuushukyuuhai", None):
                out.append((44, kyoku, False))
    return out


@pytest.mark.parametrize("always_kan", [True, False])
def test_labels_of_the_restated_loader_match_the_log(oracle, always_kan):
    O = oracle
    logs = TD._oracle_logs(O, 6, "greedy", 7000) + TD._oracle_logs(O, 2, "random", 7100)
    n_samples = n_calls = n_kan = n_riichi = n_hora = n_pass = 0
    for text in logs:
        events = [json.loads(l) for l in text.strip().splitlines()]
        for pid in range(4):
            ref = dataset_ref.load_events_by_player(O, events, pid, 3, always_include_kan_select=always_kan)
            labels = [int(x) for x in ref["actions"]]
            want = _own_actions(O, events, pid, True)
            got = [(l, ky) for l, ky in zip(labels, ref["at_kyoku"]) if l != 45]
            # without the flag the kan-select sample exists only when several kan candidates compete (gameplay.rs:353-366): it is
            # optional then, and unambiguous - all four copies of the tile are in the meld, the next discard cannot be that tile
            exp, k = [], 0
            for l, ky, sel in want:
                if sel and not always_kan and not (k < len(got) and got[k] == (l, ky)):
                    continue
                exp.append((l, ky))
                k += 1
            # the loader stops three events before the end of the log (events.windows(4)): the last actions of the game may be cut
            assert got == exp[:len(got)] and len(exp) - len(got) <= 3, (pid, got[-5:], exp[len(got) - 2:len(got) + 3])
            # a pass answers something another seat did: a discard (chi / pon / kan / ron) or a kakan (chankan)
            k = 0
            for i, cnt in dataset_ref.entry_event_indices(O, events, pid, always_kan):
                if labels[k] == 45:
                    assert cnt == 1 and events[i]["type"] in ("dahai", "kakan") and events[i]["actor"] != pid, (pid, i, events[i])
                k += cnt
            assert k == len(labels)
            assert all(bool(g) == (l <= 37) for g, l in zip(ref["apply_gamma"], labels))
            ak = list(ref["at_kyoku"])
            assert ak == sorted(ak) and all(d == (k + 1 == len(ak) or ak[k + 1] > ak[k]) for k, d in enumerate(ref["dones"]))
            n_samples += len(labels)
            n_calls += sum(1 for l in labels if 38 <= l <= 41)
            n_kan += sum(1 for l in labels if l == 42)
            n_riichi += sum(1 for l in labels if l == 37)
            n_hora += sum(1 for l in labels if l == 43)
            n_pass += sum(1 for l in labels if l == 45)
    assert n_samples > 3000 and n_calls > 20 and n_kan > 0 and n_riichi > 10 and n_hora > 10 and n_pass > 100
