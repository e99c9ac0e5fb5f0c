"""Lock-step comparison of the HIP table pool with the oracle arena on identical seeds and action streams."""
import numpy as np

KEY = 0xD5DFAA4CEF265CD7


def default_seeds(n, start=10000):
    return [(start + g // 4, KEY) for g in range(n)]


def run_lockstep(oracle, n_tables, version, max_cycles=4000, seeds=None, compare_obs=True, obs_every=1,
                 policy_seed=0x9E3779B97F4A7C15, quick_eval=True, sp_rows_checked=False, verbose=True):
    """Returns a dict with stats; raises AssertionError with a diagnostic on the first mismatch."""
    import torch

    from mortal_amd.pool import TablePool

    seeds = seeds or default_seeds(n_tables)
    arena = oracle.Arena(seeds, deal_algo=0, enable_quick_eval=quick_eval, version=version, keep_log=False)
    pool = TablePool(n_tables, version=version, deal_algo=0)
    pool.reset(seeds)
    pool.configure(0, enable_quick_eval=quick_eval)
    pool.configure(1, enable_quick_eval=quick_eval)
    C = pool.C
    # v4: rows 889.. (SP block) are produced by a separate kernel; compared only when sp_rows_checked
    n_cmp = C if (version != 4 or sp_rows_checked) else 889
    actions = None
    stats = dict(cycles=0, rows=0, obs_checked=0)
    for cycle in range(max_cycles):
        n0, n1 = pool.step(actions, None)
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
        if n == 0 and arena.n_live == 0:
            break
        obs_g, masks_g = pool.encode(0)
        want_obs = compare_obs and (cycle % obs_every == 0)
        obs_o, masks_o = arena.encode(0, n, want_obs=want_obs)
        mg = masks_g.cpu().numpy().astype(np.uint8)
        if not (mg == masks_o).all():
            r = int(np.argwhere((mg != masks_o).any(axis=1))[0][0])
            raise AssertionError(
                f"cycle {cycle}: mask mismatch at row {r} {rows_o[r]}:\n oracle {np.flatnonzero(masks_o[r]).tolist()}\n"
                f" gpu    {np.flatnonzero(mg[r]).tolist()}\n"
                f" state: {arena.player_state(int(rows_o[r][0]), int(rows_o[r][1])).snapshot()}")
        if want_obs and n:
            og = obs_g.cpu().numpy()
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
        act = oracle.random_actions(masks_o, rows_o, cycle, seed=policy_seed)
        arena.commit(act)
        actions = torch.from_numpy(act).to(pool.device)
        stats["cycles"] += 1
        stats["rows"] += n
    code, t = pool.first_error()
    assert code == 0, f"gpu table {t} in error {code}"
    scores_g, done_g = pool.results()
    cnt = pool.counters()
    scores_o = np.array([arena.result(g)[0] for g in range(n_tables)])
    done_o = np.array([arena.result(g)[1] for g in range(n_tables)])
    stats.update(counters=cnt, done_gpu=int((done_g == 1).sum()), done_oracle=int(done_o.sum()),
                 oracle_steps=int(arena.steps))
    both = (done_g == 1) & done_o
    assert (scores_g[both] == scores_o[both]).all(), "final scores differ"
    stats["scores_checked"] = int(both.sum())
    if verbose:
        print("lockstep", stats)
    pool.close()
    return stats
