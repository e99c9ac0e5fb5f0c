#!/usr/bin/env python3
"""bench.py — env steps/sec + games/sec of the MI355X table pool (BASELINE.json metric).

One "step" (cycle) = every live table advanced to its next decision point with the chosen reactions applied
(reference: one `Game::poll` + `Game::commit` round, arena/game.rs:286-304); `value` = sum over the timed cycles of
live tables / wall time = the reference's `actions` counter per second.

Default workload: 65,536 tables per GPU, uniform-random legal policy on device, obs+mask encoded for every decision
(obs version per --version), finished tables refilled with fresh seeds so the table count stays constant.
Synthetic fixed-seed deals: game g uses seed (10000 + g/4, 0xd5dfaa4cef265cd7) (docs/src/perf/strength.md:19).

  python bench.py --gpus 1 --steps 30 --warmup 10            # obs v4 (SP tables included; SP dominates the cycle)
  python bench.py --gpus 1 --steps 200 --warmup 30 --version 3  # obs v3 (no SP block): the env-step + encode path alone
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

KEY = 0xD5DFAA4CEF265CD7
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _write_ceiling_gbs(buf, nbytes):
    """The write rate this GPU sustains for a plain fill of the obs buffer (what the encode kernel's stream-out competes with),
    measured live: torch.zero_ of the same bytes, best of 5 (round 1 hard-coded one such measurement)."""
    import torch

    view = buf[: nbytes // 4]
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        view.zero_()
        e1.record()
        e1.synchronize()
        best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best

STATE_READ_BYTES = 1500  # per-decision state read
STAGGER = [True]  # --no-start-stagger: all tables start at once, like rounds 1-2 measured (the timed window then sees ONE phase)


def _cpu_worker(version, budget_s, n_tables, wid, preroll):
    """One CPU worker process: ONE oracle arena of n_tables tables under the protocol the GPU line is timed under — every slot
    parked, slot t entering play at cycle hash(t) % preroll, finished slots refilled on a fresh seed — `preroll` untimed cycles
    with masks only (no obs: the cheap way through the early game), then timed cycles with the obs of EVERY decision encoded
    (v4: SP tables included) until the budget is spent."""
    import numpy as np
    import oracle_lib as O

    O.lib()
    g0 = wid * n_tables
    seeds = [(10000 + (g0 + g) // 4, KEY) for g in range(n_tables)]
    algo = 0 if os.environ.get("MORTAL_AMD_DEAL_ALGO", "rand09").lower() in ("0", "rand08", "rand0.8", "0.8") else 1
    arena = O.Arena(seeds, deal_algo=algo, enable_quick_eval=True, version=version, keep_log=False)  # pool.default_deal_algo()
    stride = 1 << 24  # nonce stride of a restart: no other slot / worker uses the seed
    nonce = [s[0] for s in seeds]
    starts = np.zeros(n_tables, dtype=np.int64)
    if preroll > 0:
        arena.park()
        starts = np.array([(((t * 2654435761) & 0xFFFFFFFF) >> 8) % preroll for t in range(n_tables)])
    started = 0 if preroll > 0 else n_tables

    def one_cycle(cycle, want_obs):
        nonlocal started
        if cycle < preroll:
            for g in np.flatnonzero(starts == cycle).tolist():
                nonce[g] += stride
                arena.restart(g, nonce[g])
                started += 1
        rows = arena.poll()
        n = len(rows)
        if n:
            obs, masks = arena.encode(0, n, want_obs=want_obs)
            arena.commit(O.random_actions(masks, rows, cycle))
        if arena.n_live < started:  # somebody finished: restart on a fresh seed (the pool's refill mode)
            for g in range(n_tables):
                if starts[g] <= cycle and arena.result(g)[1]:
                    nonce[g] += stride
                    arena.restart(g, nonce[g])
        return n

    t_pre = time.perf_counter()
    for c in range(preroll):
        one_cycle(c, False)
    t0 = time.perf_counter()
    s0 = arena.steps
    cycle, rows_total = preroll, 0
    while time.perf_counter() - t0 < budget_s:
        rows_total += one_cycle(cycle, True)
        cycle += 1
    return dict(steps=arena.steps - s0, cycles=cycle - preroll, rows=rows_total, arenas=1, dt=time.perf_counter() - t0,
                preroll_s=t0 - t_pre, preroll_steps=s0)


def _usable_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota (containers)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, q // p))
        except (OSError, ValueError):
            pass
    return cores


PUBLISHED_CONTEXT = ("the reference publishes ONE throughput figure for this path: 'up to 40K hanchans per hour' = 11.1 hanchan/s, "
                     "Rust arena + Python net inference on an RTX 4090 + Ryzen 9 7950X, game batch size 2000 (docs/src/index.md:22,48); "
                     "env steps/s are not published (BASELINE.md section 1), so vs_baseline stays null")


def _native_oracle():
    """BASELINE.md section 3 (fallback 2): the CPU restatement compiled -O3 -march=native ON the measurement host
    (oracle/Makefile: native -> libmjoracle_native.so; never shipped).  Returns (path or None, flags string)."""
    import subprocess

    odir = os.path.join(ROOT, "oracle")
    try:
        subprocess.check_call(["make", "-C", odir, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        path = os.path.join(odir, "libmjoracle_native.so")
        # (a library that was built on another host and does not run here must not cost the line: probe it in a child process)
        ok = subprocess.call([sys.executable, "-c", f"import ctypes; ctypes.CDLL({path!r}).mjo_arena_n_live"], stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL) == 0
        if ok:
            return path, "-O3 -march=native (built on this host)"
    except (OSError, subprocess.SubprocessError):
        pass
    return None, "-O2 (portable build: the native build failed on this host)"


def cpu_baseline(version, budget_s=30.0, tables_in_flight=2000, preroll=3072):
    """Oracle arena (CPU restatement) on a bounded sample of the same workload, under the same protocol (staggered first starts
    over `preroll` untimed cycles, refill): one worker PROCESS per host core, the reference's published batch of 2,000 games in
    flight split over them (tables are independent, like the reference's rayon loop over games, arena/game.rs:286-296).
    value = sum over workers of steps_i / dt_i (all workers run concurrently)."""
    import subprocess

    cores = _usable_cores()
    n_tables = max(4, -(-tables_in_flight // cores // 4) * 4)  # whole duplicate-deal sets per worker
    lib_path, flags = _native_oracle()
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    if lib_path:
        env["MJ_ORACLE_LIB"] = lib_path
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(w), "--version", str(version),
                               "--cpu-budget", str(budget_s), "--cpu-tables", str(n_tables), "--cpu-preroll", str(preroll)],
                              stdout=subprocess.PIPE, env=env) for w in range(cores)]
    res = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0:
            res.append(json.loads(out.decode().strip().splitlines()[-1]))
    if not res:
        raise SystemExit("cpu baseline workers failed")
    value = sum(r["steps"] / r["dt"] for r in res)
    tot = {k: sum(r[k] for r in res) for k in ("steps", "cycles", "rows", "arenas")}
    import shutil

    toolchain = {"cargo": shutil.which("cargo"), "rustc": shutil.which("rustc")}  # probed on the measurement host
    return dict(value=value, unit="env steps/s", cores=len(res), kind="port", reference_toolchain=toolchain, build=flags,
                hanchan_per_hour_equiv=None, published_context=PUBLISHED_CONTEXT,
                sample=f"{len(res)} processes (1 per core) x 1 arena of {n_tables} tables = {len(res) * n_tables} games in flight (the "
                       f"reference's published batch size 2000), same protocol as the GPU line (first starts staggered over {preroll} "
                       f"untimed mask-only cycles = {sum(r['preroll_s'] for r in res) / len(res):.1f} s, finished tables refilled), then "
                       f"{budget_s:.0f} s timed: {tot['cycles']} cycles, {tot['steps']} env steps, {tot['rows']} decisions encoded "
                       f"(obs v{version}), random-legal policy, oracle {flags}; the Rust reference is not buildable here (no rustc)")


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _self_launch(n):
    """Re-exec this script under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1)."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MORTAL_AMD_BENCH_SELF_LAUNCH="1")
    return subprocess.call(cmd, env=env)


def _phase_ticks(pool):
    """Optional extra (`sp_phases`): never let it cost the benchmark line."""
    try:
        return pool.sp_phase_ticks()
    except Exception as e:  # noqa: BLE001
        print(f"sp_phase_ticks unavailable: {e}", file=sys.stderr)
        return None


def _measure(pool_cls, N, g0, world, dev, version, preroll, policy, steps, warmup, bufs, engine=None, barrier=None,
             deal_algo=None, n_pools=1):
    """One workload: N tables (as n_pools pools of N / n_pools tables, each on its own HIP stream), `preroll` untimed cycles (cheap
    v3 encode), `warmup` untimed and `steps` timed cycles of step + encode(+SP) + policy.  Returns the raw measurements (host wall
    time, HIP-event kernel times, counters).

    n_pools > 1 (--pools): the tables are INDEPENDENT, so the pool can be cut into halves that run the same cycle one after the
    other on two streams: while one half's SP kernel drains (its persistent workgroups hold every register of the chip until they
    exit), the other half's step / snapshot / encode kernels fill the CUs that fall idle.  Every table still advances once per
    cycle; `value` counts the env steps of all pools."""
    import numpy as np
    import torch

    obs, masks, act = bufs
    K = max(1, int(n_pools))
    assert N % (4 * K) == 0, "pools hold whole duplicate-deal sets"
    n = N // K
    pools, streams, views = [], [], []
    for k in range(K):
        seeds = [(10000 + (g0 + k * n + g) // 4, KEY) for g in range(n)]
        pool = pool_cls(n, version=version, deal_algo=deal_algo, device=str(dev), max_rows=2 * n)
        pool.reset(seeds, game_ids=np.arange(n), n_games_total=n)
        pool.set_refill(world * N // 4)  # a finished table restarts on a seed no other table / pool / rank uses
        if preroll > 0 and STAGGER[0]:
            # table t enters play at cycle hash(t) % preroll: after the pre-roll the pool is spread over EVERY phase of a hanchan (a
            # hanchan lasts ~3,100 cycles under the random policy).  Started all at once, the tables march in step for many
            # generations and a timed window only ever sees one phase (round 2: late south round, before the first wave of restarts)
            pool.set_start_stagger(preroll)
        pools.append(pool)
        streams.append(torch.cuda.current_stream() if K == 1 else torch.cuda.Stream(device=dev))
        C = pool.C
        r0 = 2 * n * k
        views.append(dict(obs_v=obs[r0 * C * 34: (r0 + 2 * n) * C * 34].view(2 * n, C, 34),
                          obs_3=obs[r0 * 934 * 34: (r0 + 2 * n) * 934 * 34].view(2 * n, 934, 34),
                          masks=masks[r0: r0 + 2 * n], act=act[r0: r0 + 2 * n]))
    C = pools[0].C
    use_net = [False]

    def cycle(k, i, a_prev, ob):
        pool, vw = pools[k], views[k]
        nr, _ = pool.step(a_prev, None)
        pool.encode(0, ob, vw["masks"])
        if use_net[0]:
            return engine.react_batch_device(ob[:nr], vw["masks"][:nr]), nr
        if policy == "greedy":
            pool.greedy_policy(0, vw["masks"], ob, 0x9E3779B97F4A7C15, i & 0xFFFFFFFF, vw["act"])
        else:
            pool.random_policy(0, vw["masks"], 0x9E3779B97F4A7C15, i & 0xFFFFFFFF, vw["act"])
        return vw["act"][:nr], nr

    def run(i0, i1, key, a_prev):
        rows = 0
        for i in range(i0, i1):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    a_prev[k], nr = cycle(k, i, a_prev[k], views[k][key])
                rows += nr
        return rows

    a_prev = [None] * K
    # steady-state mix of game phases: a hanchan lasts a few thousand cycles under the random policy and its kyoku end at
    # different times, so after 3072 cycles the tables are spread over every phase (SP cost depends strongly on it)
    if preroll > 0:
        for pool in pools:
            pool.configure(0, version=3)
        run(-preroll, 0, "obs_3", a_prev)
        for pool in pools:
            pool.configure(0, version=version)
    use_net[0] = engine is not None
    run(0, warmup, "obs_v", a_prev)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    c0 = [pool.counters() for pool in pools]
    ph0 = [_phase_ticks(pool) for pool in pools] if version == 4 else None
    for pool in pools:
        pool.encode_timing(True)
    t0 = time.perf_counter()
    rows_timed = run(warmup, warmup + steps, "obs_v", a_prev)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    c1 = [pool.counters() for pool in pools]
    enc_ms = enc_launches = sp_ms = sp_launches = 0
    for pool in pools:
        e_ms, e_n = pool.encode_timing(False)
        s_ms, s_n = pool.sp_timing()
        enc_ms, enc_launches, sp_ms, sp_launches = enc_ms + e_ms, enc_launches + e_n, sp_ms + s_ms, sp_launches + s_n
        code, tbl = pool.first_error()
        if code:
            raise SystemExit(f"table {tbl} in error {code}")
    ph1 = [_phase_ticks(pool) for pool in pools] if ph0 is not None else None
    def delta(key):
        return sum(b[key] - a[key] for a, b in zip(c0, c1))

    res = dict(steps=delta("steps"), games=delta("games"), dt=dt, rows=rows_timed, enc_ms=enc_ms, enc_launches=enc_launches, sp_ms=sp_ms,
               sp_launches=sp_launches, C=C, n_cycles=steps, sp_overflow=sum(c["sp_overflow"] for c in c1), n_pools=K)
    if ph0 is not None and all(p is not None for p in ph0 + ph1):
        d = {k: sum(b[k] - a[k] for a, b in zip(ph0, ph1)) for k in ph1[0]}
        tot_t = max(1, sum(d[k] for k in ("setup", "expand", "level0", "eval", "write")))
        res["sp_phases"] = {"share": {k: round(d[k] / tot_t, 4) for k in ("setup", "expand", "level0", "eval", "write")},
                            "states_per_step": d["states"] / steps, "rows_per_step": d["rows"] / steps, "overflows": d["overflow"]}
    if C == 1012:  # obs v4: the small-pool schedule of the SP kernel (rows parked by mj_k_sp and finished by mj_k_sp_wide), whole run incl. warm-up
        sc = [pool.sp_schedule_stats() for pool in pools]
        res["sp_schedule"] = {k: sum(x[k] for x in sc) for k in sc[0]}
    if world > 1:  # episode returns of EVERY pool of this rank (--pools K > 1 used to gather the first pool's only)
        parts = [pool.results() for pool in pools]
        res["results"] = (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    else:
        res["results"] = None
    for pool in pools:
        pool.close()
    return res


def _brain_workload(pool_cls, N, g0, world, dev, preroll, bufs, other_ms=None, compile_net=False):
    """BASELINE configs[2] as a driver-timed workload: the policy/value net is PyTorch's (out of this path's scope), so what the
    entry shows is the end-to-end rate and how small the environment's share of that cycle is.  compile_net: the same module under
    torch.compile (PyTorch's inductor), chunks of 8,192 rows -- the cheap settings tools/brain_tune.py found (7 x the eager forward)."""
    import torch

    from mortal_amd.policy import DeviceEngine, PolicyNet

    try:
        torch.manual_seed(0)
        engine = DeviceEngine(PolicyNet(version=4), 4, dev, enable_amp=True, compile_net=compile_net, max_batch=8192 if compile_net else 16384, min_bucket=512)
        if compile_net:
            # the two chunk shapes of this workload are compiled BEFORE the timed cycles: 8,192 rows, and the 512-row bucket of the batch's
            # ragged tail (65.8 k rows = 8 x 8,192 + 200..400).  Round 6's first line had a tail bucket seen for the first time INSIDE the 3 timed
            # cycles: ~10 s of inductor counted as 3 cycles of play (15 k instead of 145 k env steps/s)
            engine.warm(buckets=[512, 8192])
        r = _measure(pool_cls, N, g0, world, dev, 4, preroll, "brain", 3 if compile_net else 2, 1, bufs, engine)
    except Exception as e:  # noqa: BLE001 - an extra workload must never cost the headline line
        return {"error": repr(e)[:300]}
    out = _brief(r)
    # step / snapshot / row bookkeeping: what the headline's own cycle spends outside the two event-timed kernels (measured in this
    # run: ms_per_step - encode - SP of the random-policy headline, whose policy kernel is a few microseconds)
    env_ms = out["kernel_ms_per_step"]["mj_k_encode"] + out["kernel_ms_per_step"]["mj_k_sp"] + (other_ms if other_ms is not None else 0.6)
    out["env_share_of_cycle"] = env_ms / out["ms_per_step"]
    out["env_ms_per_step"] = env_ms
    out["net"] = ("random-init Brain + DQN, 192 ch x 40 blocks, torch.autocast default dtype of the backend (fp16 on ROCm, exactly "
                  "like mortal/engine.py:46), greedy" + ("; module under torch.compile (inductor), 8,192-row chunks" if compile_net else
                                                         "; eager, 16,384-row chunks (the unmodified setting)"))
    return out


def _brief(r):
    """A workload-matrix entry: the same quantities as the headline, for another workload (all driver-timed)."""
    return {"value": r["steps"] / r["dt"], "unit": "env steps/s", "ms_per_step": r["dt"] / r["n_cycles"] * 1e3,
            "games_per_sec": r["games"] / r["dt"], "decisions_per_step": r["rows"] / r["n_cycles"],
            "kernel_ms_per_step": {"mj_k_encode": r["enc_ms"] / r["n_cycles"], "mj_k_sp": r["sp_ms"] / r["n_cycles"]},
            "sp_states_per_step": (r.get("sp_phases") or {}).get("states_per_step"), "steps": r["n_cycles"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tables", type=int, default=65536, help="tables per GPU")
    ap.add_argument("--version", type=int, default=4, help="obs version (consts.rs:20-28); 4 = reference default incl. SP tables")
    ap.add_argument("--preroll", type=int, default=3072,
                    help="untimed cycles played before the warmup (with the cheap v3 encode) so that the tables are spread "
                         "over all phases of a hanchan instead of all sitting in the first turns of E1")
    ap.add_argument("--policy", choices=["random", "greedy", "brain"], default="random",
                    help="random = uniform-random legal action on device (BASELINE configs[1]); greedy = tenpai-seeking policy on "
                         "device (hands at 0..3 shanten: realistic SP load); brain = greedy argmax of a random-init network of "
                         "the reference's Brain/DQN architecture (192 ch x 40 blocks, fp16 autocast), consuming the encoded "
                         "batch in place (BASELINE configs[2])")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for --gpus N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--no-start-stagger", action="store_true",
                    help="start every table on cycle 0 (the protocol of rounds 1-2) instead of spreading the first starts over the "
                         "pre-roll (mj_pool_set_start_stagger): the timed window then shows a single phase of the hanchan")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, what the driver's N = 1, 2, 4, 8 runs measure): --tables per GPU, total work grows with N; "
                         "strong: --tables is the TOTAL over all ranks (BASELINE.md C4's '65,536 total' point), each rank owns tables / N")
    ap.add_argument("--pools", type=int, default=1,
                    help="cut the tables of a GPU into this many independent pools on their own HIP streams (same cycle, one pool after the "
                         "other): one pool's step / encode kernels run in the tail of the other's SP kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matrix", action="store_true", help="skip the extra workloads (obs v3, no pre-roll, greedy policy)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: spawn / join the --gpus N ranks, all-reduce one token over --dist-backend, print the rank "
                         "count on rank 0 and exit before any GPU work (CPU test of the launcher: tests/test_bench_contract.py)")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="timed seconds of the CPU baseline (oracle, one process per core)")
    ap.add_argument("--cpu-tables", type=int, default=2000,
                    help="CPU baseline: games in flight over all cores (the reference's published batch size); for a --cpu-worker: its own tables")
    ap.add_argument("--cpu-preroll", type=int, default=3072, help="CPU baseline: untimed mask-only cycles with staggered first starts")
    args = ap.parse_args()
    STAGGER[0] = not args.no_start_stagger
    if args.cpu_worker >= 0:
        print(json.dumps(_cpu_worker(args.version, args.cpu_budget, args.cpu_tables, args.cpu_worker, args.cpu_preroll)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a bare shell: launch the N ranks ourselves (one process per GPU), exactly the command
        # line the driver uses.  The reference covers all games from one process (arena/game.rs:286-296,
        # one_vs_three.rs:55-60); here the ranks are independent shards and rank 0 prints the whole-job line.
        sys.exit(_self_launch(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without torchrun and let it spawn the ranks)")
    if args.launch_check:
        import torch.distributed as dist

        if world > 1:
            dist.init_process_group("gloo" if args.dist_backend != "nccl" or not torch.cuda.is_available() else "nccl")
            t = torch.ones(1, dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t = t.cuda(local_rank := int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        out = {"launch_check": True, "ranks": world, "gpus": args.gpus, "self_launched": os.environ.get("MORTAL_AMD_BENCH_SELF_LAUNCH") == "1"}
        if world == 1 and args.dist_backend == "nccl" and torch.cuda.is_available():
            # one GPU: a one-rank RCCL communicator and one all-reduce on the device, so that librccl is loaded and a collective
            # has run at least once before the driver finds a multi-GPU node; no scaling claim
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                    device_id=torch.device("cuda:0"))
            t = torch.ones(4, dtype=torch.float32, device="cuda:0")
            dist.all_reduce(t)
            g = [torch.empty_like(t)]
            dist.all_gather(g, t)
            torch.cuda.synchronize()
            out["rccl_world1"] = bool((t == 1).all().item() and (g[0] == 1).all().item())
            out["rccl_loaded"] = "rccl" in open("/proc/self/maps").read()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return
    if args.dist_backend == "nccl" and world > max(1, torch.cuda.device_count()):
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node has {torch.cuda.device_count()} "
                         f"(--dist-backend gloo shares one GPU between the ranks: a smoke test of the code path, not a measurement)")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    # one rank per GPU over RCCL ("nccl"); `--dist-backend gloo` + fewer GPUs than ranks is the smoke test of this code path
    # on a single-GPU box (ranks share device 0, the three small collectives go through host tensors)
    dev_idx = local_rank % max(1, torch.cuda.device_count())
    red_dev = torch.device(f"cuda:{dev_idx}") if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_idx}"))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(dev_idx)
    dev = torch.device(f"cuda:{dev_idx}")

    from mortal_amd.pool import TablePool, default_deal_algo

    N = args.tables
    if args.scaling == "strong":
        if N % (4 * world):
            raise SystemExit(f"bench.py --scaling strong: {N} tables do not split into {world} ranks of whole duplicate-deal sets (x4)")
        N //= world
    g0 = rank * N  # tables shard by contiguous game-index ranges, multiples of 4: each duplicate-deal set stays on one GPU
    bufs = (torch.empty(2 * N * 1012 * 34, dtype=torch.float32, device=dev), torch.empty((2 * N, 46), dtype=torch.bool, device=dev),
            torch.empty(2 * N, dtype=torch.int32, device=dev))

    engine = None
    if args.policy == "brain":
        from mortal_amd.policy import DeviceEngine, PolicyNet

        torch.manual_seed(0)
        engine = DeviceEngine(PolicyNet(version=args.version if args.version >= 2 else 2), args.version, dev, enable_amp=True, max_batch=8192)  # compile_net="auto": compiled on a GPU

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    r = _measure(TablePool, N, g0, world, dev, args.version, args.preroll, args.policy, args.steps, args.warmup, bufs, engine, barrier,
                 n_pools=args.pools)
    steps, games, dt, rows_timed = r["steps"], r["games"], r["dt"], r["rows"]
    enc_ms, enc_launches, sp_ms, sp_launches, C = r["enc_ms"], r["enc_launches"], r["sp_ms"], r["sp_launches"], r["C"]

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([steps, games, rows_timed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        steps, games, rows_all = (float(x) for x in s.tolist())
        # the single collective of the data path: gather of episode returns (final scores of finished games)
        sc, dn = r["results"]
        ret = torch.from_numpy(sc).to(red_dev)
        out = [torch.empty_like(ret) for _ in range(world)] if rank == 0 else None
        dist.gather(ret, out, dst=0)
        collective_ranks = dist.get_world_size()  # the ranks the gather really spanned (RCCL when --dist-backend nccl)
    else:
        rows_all = rows_timed
        collective_ranks = 1

    # the other driver-timed workloads (N = 1 only, a few seconds each): where the headline sits between them
    matrix = None
    if world == 1 and not args.no_matrix and args.policy == "random" and args.version == 4:
        k = max(5, args.steps // 3)
        matrix = {
            "obs_v3_random": _brief(_measure(TablePool, N, g0, world, dev, 3, args.preroll, "random", 10 * k, 10, bufs)),
            "obs_v4_random_no_preroll": _brief(_measure(TablePool, N, g0, world, dev, 4, 0, "random", k, 3, bufs)),
            "obs_v4_greedy": _brief(_measure(TablePool, N, g0, world, dev, 4, args.preroll, "greedy", k, 3, bufs)),
            # BASELINE configs[1] (C2): 4,096 tables, random-action policy, env-step kernels (+ encode of every decision)
            "cfg1_4096_v3": _brief(_measure(TablePool, 4096, g0, world, dev, 3, args.preroll, "random", 20 * k, 20, bufs)),
            "cfg1_4096_v4": _brief(_measure(TablePool, 4096, g0, world, dev, 4, args.preroll, "random", 6 * k, 10, bufs)),
            # the per-GPU share of BASELINE C4's strong-scaling point (65,536 tables over 8 GPUs) and of C5 (131,072 over 8)
            "cfg_8192_v4": _brief(_measure(TablePool, 8192, g0, world, dev, 4, args.preroll, "random", 6 * k, 10, bufs)),
            "cfg_16384_v4": _brief(_measure(TablePool, 16384, g0, world, dev, 4, args.preroll, "random", 3 * k, 10, bufs)),
            "brain_v4": _brain_workload(TablePool, N, g0, world, dev, args.preroll, bufs,
                                        other_ms=(dt * 1e3 - enc_ms - sp_ms) / args.steps),
            "brain_v4_compiled": _brain_workload(TablePool, N, g0, world, dev, args.preroll, bufs,
                                                 other_ms=(dt * 1e3 - enc_ms - sp_ms) / args.steps, compile_net=True),
            "note": "cfg1_4096_v3 / _v4 = BASELINE configs[1]: 4,096 tables, uniform-random legal policy, the same protocol as the headline; "
                    "cfg_8192_v4 / cfg_16384_v4 = one GPU's share of BASELINE C4's strong-scaling point (65,536 tables over 8 GPUs) and of C5 (131,072 over 8); "
                    "brain_v4 = BASELINE configs[2]: full self-play cycle with a random-init net of the reference's Brain/DQN shape "
                    "(192 channels x 40 blocks, fp16 autocast = torch.autocast's default on this backend like mortal/engine.py:46, greedy argmax) consuming the encoded batch in place on the same GPU; "
                    "2 timed cycles (the net takes seconds per 65 k-row batch); env_share = (step + encode + SP kernels) / cycle; "
                    "brain_v4_compiled = the same module and autocast under torch.compile (PyTorch's inductor) = DeviceEngine's DEFAULT on a GPU since round 6 (compile_net='auto'; brain_v4 passes compile_net=False), 8,192-row chunks, 3 timed cycles: "
                    "the net is PyTorch's by north_star, this only shows what its cheap settings buy (tools/brain_tune.py: 25 k -> 178 k rows/s); "
                    "obs_v3_random = env-step + encode only (no SP block); obs_v4_random_no_preroll = every table in the first "
                    "turns of E1 (17 draws left: the heaviest SP phase); obs_v4_greedy = tenpai-seeking policy on device "
                    "(mj_greedy_policy; pre-rolled with the same policy): hands at 0..3 shanten, the largest SP state graphs",
        }

    if rank == 0:
        # obs v4 since round 5: mj_k_encode<4> writes rows 0..889 of a decision, the SP block (rows 889..1011, 16.6 KB) is written whole by
        # mj_k_sp (mj_encode.hip: enc_rows_written) -- the encoder's algorithmic bytes are those of the rows it writes
        enc_rows = 890 if args.version == 4 else C
        bytes_per_row = enc_rows * 34 * 4 + 46 + STATE_READ_BYTES
        achieved = rows_timed * bytes_per_row / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
        ceiling = _write_ceiling_gbs(bufs[0], int(rows_timed / max(args.steps, 1)) * enc_rows * 34 * 4)
        # HBM traffic of the encode kernel from the separate rocprofv3 --pmc passes (tools/profile_round.sh ->
        # profiles/pmc_encode.json: WRITE_SIZE + 2 x FETCH_SIZE per decision), scaled to this run's rows per launch
        traffic = traffic_src = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_encode.json")
        if os.path.exists(pmc_file) and enc_launches and args.version == 4:
            pmc = json.load(open(pmc_file))
            if "write_bytes_per_decision" in pmc and "fetch_bytes_per_decision" in pmc:
                per = pmc["write_bytes_per_decision"] + pmc["fetch_bytes_per_decision"]
                traffic = per * rows_timed / enc_launches
                traffic_src = (f"profiles/pmc_encode.json ({per:.0f} B/decision; static file from separate rocprofv3 --pmc passes at "
                               f"{pmc.get('measured_at_tables', pmc['tables'])} tables, {pmc.get('measured_utc', 'round 2')})")
        line = {
            "metric": ("env steps/sec (65536 parallel tables per GPU)" if args.scaling == "weak"
                       else f"env steps/sec ({args.tables} parallel tables in total)"),
            "value": steps / dt,
            "unit": "env steps/s",
            "n_gpus": world,
            "ranks": world,
            "dist_backend": (args.dist_backend if world > 1 else None),
            "collective_ranks": collective_ranks,  # dist.get_world_size() after the gather of episode returns ("rccl_ranks" over nccl)
            "rccl_ranks": (collective_ranks if world > 1 and args.dist_backend == "nccl" else None),
            "gpus_visible_per_rank": torch.cuda.device_count(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8/i32 state, f32 obs",
            "data": "synthetic",
            "games_per_sec": games / dt,
            "decisions_per_sec": rows_all / dt,
            "config": {
                "workload": f"{N} tables per GPU" + (f" ({args.tables} in total, strong scaling)" if args.scaling == "strong" else "") + ", "
                            + {"random": "uniform-random legal policy on device", "greedy": "tenpai-seeking policy on device",
                               "brain": "greedy policy of a random-init Brain/DQN-shaped net (192x40, fp16 autocast) on the same device"}[args.policy]
                            + f", env-step + obs(v{args.version})"
                            f"+mask encode of every decision, finished tables refilled; fixed-seed synthetic deals "
                            f"(wall shuffle of rand {'0.9.1' if default_deal_algo() else '0.8'}); "
                            f"{args.preroll} untimed pre-roll cycles"
                            + (", first starts staggered over them: every phase of a hanchan is present" if STAGGER[0] and args.preroll > 0
                               else ", all tables started together: one phase of the hanchan"),
                "preroll_cycles": args.preroll,
                "start_stagger": bool(STAGGER[0] and args.preroll > 0),
                "policy": args.policy,
                "tables_per_gpu": N,
                "pools_per_gpu": args.pools,
                "obs_version": args.version,
                "parallelism": f"tables sharded x{world}, no data-path collective (one RCCL gather of episode returns)",
            },
            "roofline": {
                "kernel": "mj_k_encode",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_write_ceiling": achieved / ceiling,
                "measured_write_ceiling": ceiling,  # torch zero_ of one launch's obs bytes on this GPU, measured in this run
                "traffic": traffic,
                "traffic_unit": "B/launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_per_row * rows_timed / max(enc_launches, 1),
                "bytes_per_decision": bytes_per_row,
                "obs_rows_written_per_decision": enc_rows,
                "avg_launch_ms": enc_ms / max(enc_launches, 1),
                "launches": enc_launches,
            },
            # share of the timed wall clock spent in each timed kernel (HIP events); for obs v4 the SP-table kernel
            # (single-player win/tenpai probability tables, obs rows 889..1011) dominates the cycle
            "kernel_ms_per_step": {"mj_k_encode": enc_ms / args.steps, "mj_k_sp": sp_ms / args.steps,
                                   "everything_else": (dt * 1e3 - enc_ms - sp_ms) / args.steps},
        }
        if "sp_phases" in r:  # where mj_k_sp spends its workgroup time (shares of the summed phase timers) + states per cycle
            line["sp_phases"] = r["sp_phases"]
            line["roofline_sp"] = _roofline_sp(r, sp_ms, sp_launches)
        if "sp_schedule" in r:
            line["sp_schedule"] = r["sp_schedule"]
        if matrix:
            line["workloads"] = matrix
        if not args.no_cpu_baseline and world == 1 and args.policy == "random":  # reported at N=1 only (rank 0)
            cb = cpu_baseline(args.version, args.cpu_budget, args.cpu_tables, args.cpu_preroll)
            # the published figure's unit, for the eye only: hanchan/h at the measured games-per-step ratio of the GPU run
            cb["hanchan_per_hour_equiv"] = cb["value"] * (games / max(steps, 1)) * 3600.0
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md: v_fma_f32)
# What this kernel's instruction mix can issue at best: tools/ubench_valu.hip (profiles/r03_ubench_valu.jsonl, MI355X, 4 waves
# per SIMD) measures 4.2-4.4 cycles per wave-instruction per SIMD for v_bfe / v_lshrrev_b64 / v_mul_lo / v_mad_u32_u24 / v_bcnt /
# v_min3 / v_perm / v_lshl_add / v_cmp / DPP moves / v_pk_fma_f32, 3.45 for v_fma_f32 and 2.4-2.6 only for v_add_u32 / v_and_b32 /
# v_mul_f32 / v_add_f32 - the integer / bit-field work of mj_k_sp is priced at 4.3.
VALU_PEAK_WAVE_INSTS_INT_MIX = 256 * 4 * 2.4e9 / 4.3
def _latest_sp_pmc():
    """profiles/rNN_sp_pmc.json of the latest round (tools/pmc_sp.sh + tools/summarize_sp_pmc.py write one per round)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_sp_pmc.json")))
    return files[-1] if files else os.path.join(ROOT, "profiles", "r03_sp_pmc.json")


SP_PMC_FILE = _latest_sp_pmc()


def sp_source_sha16():
    """Identity of the SP kernel's sources: the PMC summary is stamped with it, and a bench line refuses to quote instruction
    counts measured on different code (stale static profiles must not look measured)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("mj_sp.hip", "mj_sptab.h", "mj_algo.h", "mj_rules.h", "mj_state.h"):
        with open(os.path.join(ROOT, "mortal_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _roofline_sp(r, sp_ms, sp_launches):
    """The cycle's dominant kernel is neither HBM- nor MFMA-bound: integer / bit-field VALU work (table-id shanten sets, hashing,
    child lists) and exact f32 sums in the reference's order, much of it waiting for dependent gathers.  Model: VALU
    wave-instructions per state-graph node (separate rocprofv3 --pmc pass, profiles/rNN_sp_pmc.json - a STATIC file, stamped with
    the hash of the kernel sources; if it does not match the sources of this run the fraction is null and `stale` is set)
    x nodes visited in the timed region (counted by the kernel) / kernel time (HIP events on the launch stream), against the
    guide's VALU issue peak (`peak` / `frac`: comparable across rounds) and, separately, against the issue rate the instruction mix
    reached in the builder's micro-benchmark (`frac_of_measured_int_issue`).  `states_per_sec` is the work rate."""
    states = r["sp_phases"]["states_per_step"] * r["n_cycles"]
    out = {"kernel": "mj_k_sp", "bound": "valu", "unit": "G wave-instructions/s", "peak": VALU_PEAK_WAVE_INSTS / 1e9,
           "peak_basis": "MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD x 1,024 SIMDs x 2.4 GHz (the same "
                         "denominator as rounds 1-2; round 3's line priced against peak_measured_int_issue instead)",
           "peak_measured_int_issue": VALU_PEAK_WAVE_INSTS_INT_MIX / 1e9,
           "peak_measured_int_issue_basis": "tools/ubench_valu.hip, 4 waves per SIMD: 4.3 cycles per wave-instruction for the kernel's "
                                            "integer / bit-field classes (profiles/r03_ubench_valu.jsonl; builder-run)",
           # the work rate: the number that has to rise from round to round whatever the instruction count does
           "states_per_sec": states / (sp_ms * 1e-3) if sp_ms > 0 else None,
           "avg_launch_ms": sp_ms / max(sp_launches, 1), "launches": sp_launches,
           "states_per_launch": r["sp_phases"]["states_per_step"], "from_static_profile": True}
    if os.path.exists(SP_PMC_FILE) and sp_ms > 0:
        pmc = json.load(open(SP_PMC_FILE))
        stale = pmc.get("source_sha16") != sp_source_sha16()
        out["stale"] = stale
        out["static_profile"] = {"file": os.path.relpath(SP_PMC_FILE, ROOT), "measured_utc": pmc.get("measured_utc"),
                                 "measured_at_tables": pmc.get("measured_at_tables"), "source_sha16": pmc.get("source_sha16")}
        insts = pmc["valu_insts_per_state"] * r["sp_phases"]["states_per_step"] * r["n_cycles"]
        out["valu_insts_per_state"] = pmc["valu_insts_per_state"]
        out["achieved"] = None if stale else insts / (sp_ms * 1e-3) / 1e9
        out["frac"] = None if stale else out["achieved"] / out["peak"]
        out["frac_of_measured_int_issue"] = None if stale else out["achieved"] / out["peak_measured_int_issue"]
        for k in ("valu_busy", "lane_utilisation", "wave_wait_share", "hbm_fetch_bytes_per_state", "hbm_write_bytes_per_state",
                  "l2_hit_rate", "salu_insts_per_state", "lds_insts_per_state", "vmem_rd_insts_per_state", "source"):
            if k in pmc:
                out[k] = pmc[k]
        if "hbm_fetch_bytes_per_state" in pmc and not stale:
            by = (pmc["hbm_fetch_bytes_per_state"] + pmc["hbm_write_bytes_per_state"]) * r["sp_phases"]["states_per_step"] * r["n_cycles"]
            out["hbm_gbs"] = by / (sp_ms * 1e-3) / 1e9
            out["hbm_frac_of_peak"] = out["hbm_gbs"] / HBM_PEAK_GBS
    return out


if __name__ == "__main__":
    main()
