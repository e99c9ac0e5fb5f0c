"""Lock-step comparison of the HIP table pool with the oracle arena on identical seeds and action streams."""
import numpy as np
import torch

from mortal_amd.pool import OBS_ROWS

KEY = 0xD5DFAA4CEF265CD7


def default_seeds(n, start=10000):
    return [(start + g // 4, KEY) for g in range(n)]


DISCARD_ROW = {1: 923, 2: 927, 3: 919, 4: 874}  # first row of the 5-row discard block (obs_repr.rs:431-476)


def greedy_actions(masks, rows, cycle, obs_rows3, seed):
    """A test policy that plays towards tenpai so that riichi / ron / tsumo / furiten / SP paths are exercised:
    always agari, usually riichi, discards that lower (else keep) the shanten number, occasional calls.
    obs_rows3 = the [n, 3, 34] slice (discard candidates, keep-shanten, next-shanten rows) of the encoded obs."""
    masks = np.asarray(masks, dtype=bool)
    n = len(masks)
    act = np.full(n, 45, dtype=np.int32)
    if n == 0:
        return act
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) ^ (rows[:, 0] * np.uint64(0xD1B54A32D192ED03)) ^ (rows[:, 1] * np.uint64(0x8CB92BA72F3D8DD7))
             ^ (rows[:, 2] * np.uint64(0xAEF17502108EF2D9)) ^ (np.array([cycle], dtype=np.uint64) * np.uint64(0x94D049BB133111EB)))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    h = (x >> np.uint64(20)).astype(np.int64)
    deaka = np.arange(37)
    deaka[34:] = [4, 13, 22]

    def pick(cands, hv):
        idx = np.flatnonzero(cands)
        return int(idx[hv % len(idx)])

    for r in range(n):
        m, hv = masks[r], int(h[r])
        if rows[r, 2]:  # kan-select row: any legal tile
            act[r] = pick(m, hv)
            continue
        if m[43]:
            act[r] = 43
        elif m[37] and hv % 8 != 0:
            act[r] = 37
        elif m[44] and hv % 2 == 0:
            act[r] = 44
        elif m[42] and m[:37].any() and hv % 3 == 0:
            act[r] = 42
        elif m[:37].any():
            legal = m[:37]
            nxt = legal & (obs_rows3[r, 2][deaka] > 0)
            keep = legal & (obs_rows3[r, 1][deaka] > 0)
            act[r] = pick(nxt if nxt.any() else keep if keep.any() else legal, hv // 7)
        else:  # reaction to a discard
            if m[41] and hv % 3 == 0:
                act[r] = 41
            elif m[38:41].any() and hv % 4 == 0:
                act[r] = 38 + pick(m[38:41], hv // 5)
            elif m[42] and hv % 2 == 0:
                act[r] = 42
            else:
                act[r] = 45
        assert m[act[r]], (r, act[r])
    return act


def tsumogiri_actions(arena, masks, rows):
    """agent/tsumogiri.rs:17-38: discard the tile just drawn whenever a discard is possible, otherwise pass."""
    masks = np.asarray(masks, dtype=bool)
    act = np.full(len(masks), 45, dtype=np.int32)
    for r in range(len(masks)):
        if rows[r][2]:  # kan-select rows never occur (no kan is ever declared)
            raise AssertionError("kan-select row under the tsumogiri policy")
        if masks[r, :37].any():
            act[r] = arena.player_state(int(rows[r][0]), int(rows[r][1])).snapshot()["last_self_tsumo"]
        assert masks[r, act[r]], (r, int(act[r]), np.flatnonzero(masks[r]).tolist())
    return act


def fake_q_values(masks, rows, cycle, seed):
    """Deterministic stand-in for the engine's q-values: a hash per (row, action) in [-1, 1), -inf where illegal
    (engine.py masks illegal actions the same way), with deliberate ties to exercise max_by's last-maximum rule."""
    masks = np.asarray(masks, dtype=bool)
    n = len(masks)
    rows = np.asarray(rows, dtype=np.uint64).reshape(n, 3)
    with np.errstate(over="ignore"):
        base = (np.uint64(seed) ^ (rows[:, 0] * np.uint64(0xD1B54A32D192ED03)) ^ (rows[:, 1] * np.uint64(0x8CB92BA72F3D8DD7))
                ^ (np.array([cycle], dtype=np.uint64) * np.uint64(0x94D049BB133111EB)))
        x = base[:, None] + np.arange(46, dtype=np.uint64)[None, :] * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    q = ((x >> np.uint64(60)).astype(np.float32) - 8.0) / 8.0  # 16 levels => frequent ties
    q[~masks] = -np.inf
    return np.ascontiguousarray(q, dtype=np.float32)


def run_lockstep(oracle, n_tables, version, max_cycles=4000, seeds=None, compare_obs=True, obs_every=1,
                 policy_seed=0x9E3779B97F4A7C15, quick_eval=True, sp_rows_checked=False, verbose=True,
                 policy="random", guard=False, oracle_obs=False, compare_logs=False, deal_algo=0, refill=0, min_games=2,
                 threads=0, obs_cycles=None, pool_cls=None, stagger=0, obs_slice=0, obs_slice_max=0):
    """Returns a dict with stats; raises AssertionError with a diagnostic on the first mismatch.

    refill = nonce stride: finished slots restart on (nonce + stride, key) on both sides (the pool's steady-state mode,
    mj_pool_set_refill / mj_k_refill — what bench.py times) until every slot has finished `min_games` hanchan.
    stagger = S (needs refill): the protocol bench.py times — every slot is parked before the first cycle and slot t enters play
    at cycle ((t * 2654435761 mod 2^32) >> 8) % S through the refill path (mj_pool_set_start_stagger, mj_step.hip: mj_k_park /
    mj_k_refill), i.e. on (nonce + stride, key) as game id t + n_tables; the oracle slot idles until that cycle the same way.
    obs_cycles: explicit set of cycles whose obs are compared (overrides obs_every); threads: oracle encode threads;
    obs_slice: compare the obs in slices of that many rows (the oracle encodes slice by slice: the 65,536-table pool's 8 GB
    batch never exists twice on the host); obs_slice_max = k > 0: only k of those slices, spread evenly over the batch (rows are
    ordered by table, and a table's phase does not depend on its index: any slice holds the batch's mix of decisions)."""
    import torch

    if pool_cls is None:
        from mortal_amd.pool import TablePool as pool_cls  # tests/test_emu_*.py pass the host emulation of the same kernels

    seeds = seeds or default_seeds(n_tables)
    arena = oracle.Arena(seeds, deal_algo=deal_algo, enable_quick_eval=quick_eval, version=version, keep_log=compare_logs)
    pool = pool_cls(n_tables, version=version, deal_algo=deal_algo)
    if compare_logs:
        pool.enable_log()
    gens_total = 16 if refill else 1
    pool.reset(seeds, game_ids=np.arange(n_tables), n_games_total=gens_total * n_tables)
    if refill:
        pool.set_refill(refill)
    gen = [0] * n_tables
    nonce = [int(s[0]) for s in seeds]
    starts = None
    if stagger:
        assert refill, "the staggered start goes through the refill path"
        pool.set_start_stagger(stagger)
        arena.park()
        starts = [(((t * 2654435761) & 0xFFFFFFFF) >> 8) % stagger for t in range(n_tables)]
        starts_np = np.array(starts)
    gen_scores = {}
    pool.configure(0, enable_quick_eval=quick_eval, enable_rule_based_agari_guard=guard)
    pool.configure(1, enable_quick_eval=quick_eval, enable_rule_based_agari_guard=guard)
    C = pool.C
    # v4: rows 889.. (SP block) are produced by a separate kernel; compared only when sp_rows_checked
    n_cmp = C if (version != 4 or sp_rows_checked) else 889
    actions = q_dev = None
    stats = dict(cycles=0, rows=0, obs_checked=0)
    obs_buf = [None, None]  # the device's obs / mask buffers, reused across cycles
    for cycle in range(max_cycles):
        if starts is not None:
            for g in np.flatnonzero(starts_np == cycle).tolist():
                if starts[g] == cycle:  # the device's refill kernel runs ahead of this cycle's step: in play from this cycle on
                    gen[g] = 1
                    nonce[g] += refill
                    arena.restart(g, nonce[g])
        n0, n1 = pool.step(actions, None, q_dev, None)
        assert n1 == 0
        rows_o = arena.poll()
        rows_g = pool.rows(0)
        if len(rows_o) != len(rows_g) or (len(rows_o) and not (rows_o == rows_g).all()):
            k = 0
            m = min(len(rows_o), len(rows_g))
            while k < m and (rows_o[k] == rows_g[k]).all():
                k += 1
            msg = [f"cycle {cycle}: row lists differ at index {k}: oracle {rows_o[k] if k < len(rows_o) else None} "
                   f"gpu {rows_g[k] if k < len(rows_g) else None} (n_oracle={len(rows_o)}, n_gpu={len(rows_g)})"]
            bad_game = int(rows_o[k][0] if k < len(rows_o) else rows_g[k][0])
            msg.append(f"oracle view of game {bad_game}: {arena.game_view(bad_game).tolist()}")
            for s in range(4):
                ps = arena.player_state(bad_game, s)
                if ps is not None:
                    sn = ps.snapshot()
                    msg.append(f" seat {s}: cans={sn['cans']} shanten={sn['shanten']} tehai={sn['tehai'].tolist()}")
            code, t = pool.first_error()
            msg.append(f"gpu first error: code {code} table {t}")
            raise AssertionError("\n".join(msg))
        n = len(rows_o)
        if n == 0 and arena.n_live == 0 and not (starts is not None and cycle < stagger):
            break
        # buffers poisoned before the encode: every cell of the observation has to be WRITTEN by a kernel (the encoder owns rows
        # 0..888 of obs v4 and mj_k_sp the rest), nothing may rely on what the allocator handed out
        n_g = pool.n_rows[0]
        # (ONE buffer that grows geometrically: a fresh torch.full of a slightly larger size every cycle of a staggered start makes torch's
        # caching allocator keep ~3,000 blocks of distinct sizes -- it fills the 288 GB, and the HIP runtime's own next allocation,
        # e.g. the scratch of a kernel's first launch, aborts the process with HSA_STATUS_ERROR_OUT_OF_RESOURCES)
        C_g = OBS_ROWS[pool.versions[0]]
        if obs_buf[0] is None or obs_buf[0].shape[0] < n_g:
            obs_buf[0] = obs_buf[1] = None
            cap_g = max(n_g + n_g // 4, 64)
            obs_buf[0] = torch.empty((cap_g, C_g, 34), dtype=torch.float32, device=pool.device)
            obs_buf[1] = torch.empty((cap_g, 46), dtype=torch.bool, device=pool.device)
        obs_g = obs_buf[0][:n_g].fill_(float("nan"))
        masks_g = obs_buf[1][:n_g].fill_(True)
        obs_g, masks_g = pool.encode(0, obs_g, masks_g)
        want_obs = compare_obs and ((cycle in obs_cycles) if obs_cycles is not None else (cycle % obs_every == 0))
        sliced = bool(want_obs and obs_slice and n > obs_slice)
        obs_o, masks_o = arena.encode(0, n, want_obs=want_obs and not sliced, threads=threads)
        mg = masks_g.cpu().numpy().astype(np.uint8)
        if not (mg == masks_o).all():
            r = int(np.argwhere((mg != masks_o).any(axis=1))[0][0])
            raise AssertionError(
                f"cycle {cycle}: mask mismatch at row {r} {rows_o[r]}:\n oracle {np.flatnonzero(masks_o[r]).tolist()}\n"
                f" gpu    {np.flatnonzero(mg[r]).tolist()}\n"
                f" state: {arena.player_state(int(rows_o[r][0]), int(rows_o[r][1])).snapshot()}")
        if n and not bool(masks_g.any(dim=1).all()):
            raise AssertionError(f"cycle {cycle}: a decision row without any legal action")
        if sliced:
            firsts = list(range(0, n, obs_slice))
            if obs_slice_max and len(firsts) > obs_slice_max:
                firsts = [firsts[(k * (len(firsts) - 1)) // max(1, obs_slice_max - 1)] for k in range(obs_slice_max)]
                firsts = sorted({max(0, min(r0, n - obs_slice)) for r0 in firsts})  # whole slices (the batch's last one is ragged)
            for r0 in firsts:
                r1 = min(n, r0 + obs_slice)
                og = obs_g[r0:r1].cpu().numpy()
                oo, _ = arena.encode(r0, r1, want_obs=True, threads=threads)
                if not (np.isfinite(og).all() and og.min() >= 0.0 and og.max() <= 1.0):
                    raise AssertionError(f"cycle {cycle}: obs value outside [0, 1] in rows {r0}..{r1}")
                a = np.ascontiguousarray(og[:, :n_cmp]).view(np.uint32)
                b = np.ascontiguousarray(oo[:, :n_cmp]).view(np.uint32)
                if not (a == b).all():
                    bad = np.argwhere(a != b)
                    r = int(bad[0][0])
                    raise AssertionError(f"cycle {cycle}: obs mismatch at row {r0 + r} {rows_o[r0 + r]}; differing obs rows "
                                         f"{sorted(set(int(x[1]) for x in bad if x[0] == r))[:40]}")
                stats["obs_checked"] += r1 - r0
        elif want_obs and n:
            og = obs_g.cpu().numpy()
            # obs_repr.rs:626-628: every plane value lies in [0, 1] (the reference's debug assertion on the finished tensor)
            if not (np.isfinite(og).all() and og.min() >= 0.0 and og.max() <= 1.0):
                raise AssertionError(f"cycle {cycle}: obs value outside [0, 1]: min {og.min()} max {og.max()}")
            # state/test.rs:49-58 (`validate` after every event): shanten bookkeeping of the acting seat is consistent with a
            # fresh table lookup of its hand, on the oracle's PlayerState (the device agrees with it field by field below)
            for r in range(0, n, max(1, n // 32)):
                sn = arena.player_state(int(rows_o[r][0]), int(rows_o[r][1])).snapshot()
                assert sn["real_time_shanten"] == oracle.calc_shanten(sn["tehai"], sn["tehai_len_div3"]), (cycle, rows_o[r].tolist())
                assert sn["doras_owned"][0] >= int(sn["akas_in_hand"].sum())
            a = np.ascontiguousarray(og[:, :n_cmp]).view(np.uint32)
            b = np.ascontiguousarray(obs_o[:, :n_cmp]).view(np.uint32)
            if not (a == b).all():
                bad = np.argwhere(a != b)
                r = int(bad[0][0])
                rows_bad = sorted(set(int(x[1]) for x in bad if x[0] == r))
                lines = [f"cycle {cycle}: obs mismatch, {len(set(int(x[0]) for x in bad))} of {n} rows differ; "
                         f"first row {r} {rows_o[r]}; differing obs rows {rows_bad[:40]}"]
                for rr in rows_bad[:12]:
                    lines.append(f"  obs row {rr}: oracle {obs_o[r, rr].tolist()}\n           gpu    {og[r, rr].tolist()}")
                hist = {}
                for x in bad:
                    hist[int(x[1])] = hist.get(int(x[1]), 0) + 1
                lines.append(f"  histogram of differing obs rows over the batch: {sorted(hist.items())[:60]}")
                g, seat = int(rows_o[r][0]), int(rows_o[r][1])
                dbg = pool.debug_table(g)
                ps = arena.player_state(g, seat)
                oya = int(dbg["kyoku"][0]) & 3
                for rel in range(4):
                    a = (seat + rel) & 3
                    kg = dbg["kawa"].reshape(4, -1)[a][: int(dbg["kawa_len"][a])]
                    pad = 1 if rel < ((oya - seat) & 3) else 0
                    lines.append(f"  kawa rel {rel} (abs {a}) pad {pad}: gpu    {[hex(int(x)) for x in kg]}")
                    lines.append(f"                              oracle {[hex(int(x)) for x in ps.kawa(rel)]}")
                lines.append(f"  oracle snapshot: {ps.snapshot()}")
                lines.append(f"  gpu table: " + ", ".join(f"{k}={v.tolist()}" for k, v in dbg.items()
                                                         if k not in ("wall", "kawa")))
                raise AssertionError("\n".join(lines))
            stats["obs_checked"] += n
        if oracle_obs and n and cycle % obs_every == 0:
            inv_g = pool.encode_oracle(0).cpu().numpy()
            inv_o = arena.encode_oracle(0, n, version)
            if not (inv_g.view(np.uint32) == inv_o.view(np.uint32)).all():
                bad = np.argwhere(inv_g.view(np.uint32) != inv_o.view(np.uint32))
                r = int(bad[0][0])
                rr = sorted(set(int(x[1]) for x in bad if x[0] == r))
                lines = [f"cycle {cycle}: invisible obs mismatch at row {r} {rows_o[r]}; differing planes {rr[:40]}"]
                for q in rr[:8]:
                    lines.append(f"  plane {q}: oracle {inv_o[r, q].tolist()}\n            gpu    {inv_g[r, q].tolist()}")
                raise AssertionError("\n".join(lines))
            stats["oracle_obs_checked"] = stats.get("oracle_obs_checked", 0) + n
        if policy == "greedy":
            d0 = DISCARD_ROW[version]
            act = greedy_actions(masks_o, rows_o, cycle, obs_g[:, d0:d0 + 3].cpu().numpy(), policy_seed)
            if n:  # the device-side port of the same policy (mj_k_greedy_policy, the benchmark's realistic-hand workload)
                act_dev = pool.greedy_policy(0, masks_g, obs_g, policy_seed, cycle).cpu().numpy()
                assert (act_dev == act).all(), f"cycle {cycle}: device greedy policy differs at rows {np.flatnonzero(act_dev != act)[:8].tolist()}"
        elif policy == "tsumogiri":
            act = tsumogiri_actions(arena, masks_o, rows_o)
        else:
            act = oracle.random_actions(masks_o, rows_o, cycle, seed=policy_seed)
        if guard:
            q = fake_q_values(masks_o, rows_o, cycle, policy_seed)
            arena.commit(act, q)
            q_dev = torch.from_numpy(q).to(pool.device)
        else:
            arena.commit(act)
        actions = torch.from_numpy(act).to(pool.device)
        stats["cycles"] += 1
        stats["rows"] += n
        if refill:
            # (a slot can only have finished if fewer games are live than have entered play: big pools skip the per-slot scan on the
            # many cycles in which nobody finishes)
            started = n_tables if starts is None else int((starts_np <= cycle).sum())
            fin = ()
            if arena.n_live < started:
                dflags = arena.done_flags() != 0
                fin = np.flatnonzero(dflags if starts is None else dflags & (starts_np <= cycle)).tolist()
            for g in fin:
                sc, dn = arena.result(g)
                if dn and (starts is None or starts[g] <= cycle):  # a parked slot is "finished" without having played
                    gen_scores[(g, gen[g])] = sc.copy()
                    gen[g] += 1
                    nonce[g] += refill
                    arena.restart(g, nonce[g])
            if min(gen) >= min_games + (1 if stagger else 0):
                break
    code, t = pool.first_error()
    assert code == 0, f"gpu table {t} in error {code}"
    scores_g, done_g = pool.results()
    cnt = pool.counters()
    stats["sp_schedule"] = pool.sp_schedule_stats()  # (small-pool schedule of mj_k_sp: rows parked / finished by mj_k_sp_wide)
    if refill:
        # every hanchan the oracle finished (slot g, generation k) is game id g + k * n_tables on the device
        assert cnt["steps"] == arena.steps, (cnt["steps"], arena.steps)
        for (g, k), sc in gen_scores.items():
            if k < gens_total:
                gid = g + k * n_tables
                assert done_g[gid] == 1 and (scores_g[gid] == sc).all(), f"slot {g} generation {k}: {scores_g[gid]} vs {sc}"
        stats.update(counters=cnt, games_checked=len(gen_scores), generations=(min(gen), max(gen)), oracle_steps=int(arena.steps),
                     scores_checked=len(gen_scores), done_gpu=int((done_g == 1).sum()), done_oracle=len(gen_scores),
                     guard_hits=int(arena.guard_hits))
        if verbose:
            print("lockstep refill", stats)
        pool.close()
        return stats
    scores_o = np.array([arena.result(g)[0] for g in range(n_tables)])
    done_o = np.array([arena.result(g)[1] for g in range(n_tables)])
    stats.update(counters=cnt, done_gpu=int((done_g == 1).sum()), done_oracle=int(done_o.sum()),
                 oracle_steps=int(arena.steps), guard_hits=int(arena.guard_hits))
    both = (done_g[:n_tables] == 1) & done_o
    assert (scores_g[:n_tables][both] == scores_o[both]).all(), "final scores differ"
    stats["scores_checked"] = int(both.sum())
    if compare_logs:
        import json

        from mortal_amd import mjai_log

        n_ev = 0
        for g, words in enumerate(pool.read_logs()):
            got = mjai_log.decode_events(words)
            want = arena.log(g)
            dump = lambda e: json.dumps(e, separators=(",", ":"))
            if [dump(e) for e in got] != [dump(e) for e in want]:
                k = 0
                while k < min(len(got), len(want)) and dump(got[k]) == dump(want[k]):
                    k += 1
                raise AssertionError(f"game {g}: event {k} of {len(want)} differs:\n oracle {want[k] if k < len(want) else None}\n"
                                     f" gpu    {got[k] if k < len(got) else None}")
            n_ev += len(got)
        stats["log_events_checked"] = n_ev
    if verbose:
        print("lockstep", stats)
    pool.close()
    return stats
