"""The driver's bench.py contract, checked on the committed line of the last builder GPU run (profiles/r04_bench_v4.json), on the
driver's own records (BENCH_r0N.json, when present) and on bench.py's argument defaults — no GPU needed.  Guards against drift
between the JSON the driver parses, BASELINE.json's metric, and the numbers quoted in DESIGN.md §7."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_v4.json")))  # the latest round's committed line
    with open(files[-1]) as f:
        return json.load(f)


def _driver_lines():
    """(round, parsed bench line) of every BENCH_r0N.json the driver left in the repository root."""
    import glob

    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json"))):
        with open(f) as fh:
            d = json.load(fh)
        p = d.get("parsed") or d
        if isinstance(p, dict) and "value" in p:
            out.append((os.path.basename(f)[6:9], p))
    return out


def test_committed_bench_line_has_the_contract_keys_and_consistent_arithmetic():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["unit"] == "env steps/s" and d["metric"].startswith("env steps/sec") and base["metric"].startswith("env steps/sec")
    assert "65536" in d["metric"] and "65536" in base["metric"] and base["published"] == {}  # nothing published => vs_baseline null
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = env steps of the timed region / its duration; a cycle advances every live table by one step
    tables = d["config"]["tables_per_gpu"] * d["n_gpus"]
    steps_per_cycle = d["value"] * d["ms_per_step"] / 1e3
    assert 0.95 * tables <= steps_per_cycle <= tables  # finished tables are refilled, a few are between games
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2  # no wasted re-reads / rewrites
    s = d["roofline_sp"]
    assert s["bound"] == "valu" and s["from_static_profile"] is True and s["stale"] is False  # the PMC summary matches the kernel sources
    # `frac` is priced against the guide's VALU issue peak (comparable across rounds); the builder's micro-benchmark rate is a separate key
    assert abs(s["frac"] - s["achieved"] / s["peak"]) < 1e-6 and abs(s["peak"] - 1228.8) < 0.01 and s["peak_measured_int_issue"] < s["peak"]
    assert abs(s["frac_of_measured_int_issue"] - s["achieved"] / s["peak_measured_int_issue"]) < 1e-6
    assert abs(s["states_per_sec"] - s["states_per_launch"] / (s["avg_launch_ms"] * 1e-3)) / s["states_per_sec"] < 1e-3
    assert abs(s["achieved"] - s["valu_insts_per_state"] * s["states_per_launch"] / (s["avg_launch_ms"] * 1e-3) / 1e9) / s["achieved"] < 1e-3
    assert s["static_profile"]["measured_at_tables"] == 65536 and "65536 tables" in r["traffic_source"]
    k = d["kernel_ms_per_step"]
    assert abs(sum(k.values()) - d["ms_per_step"]) / d["ms_per_step"] < 0.02  # the per-kernel split covers the cycle
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0
    for w in ("obs_v3_random", "obs_v4_random_no_preroll", "obs_v4_greedy", "cfg1_4096_v3", "cfg1_4096_v4", "brain_v4", "brain_v4_compiled"):
        assert d["workloads"][w]["value"] > 0
    assert d["workloads"]["brain_v4"]["env_share_of_cycle"] < 0.05  # BASELINE configs[2]: the net, not the environment, is the cycle
    assert d["steps"] >= 100 and d["config"]["start_stagger"] is True


def test_design_quotes_the_committed_numbers():
    d = _line()
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        text = f.read()
    head = f"{d['value'] / 1e6:.2f} M env steps/s"
    assert head in text, head
    assert f"{d['ms_per_step']:.1f} ms/cycle" in text
    assert f"`mj_k_sp` {d['kernel_ms_per_step']['mj_k_sp']:.1f} ms" in text
    assert "builder-run" in text  # the committed line is the builder's box; the numbers of record are the driver's
    # ... and the driver's last record is quoted next to it (DESIGN quoted the builder's best box only)
    drv = [(r, p) for r, p in _driver_lines() if r == "r05"]  # the last driver record when this text was written
    if drv:
        p = drv[0][1]
        assert f"{p['value'] / 1e6:.3f} M env steps/s" in text and f"{p['ms_per_step']:.1f} ms/cycle" in text


def test_driver_records_are_consistent_with_the_contract():
    """Every BENCH_r0N.json the driver wrote: headline keys, the metric, the roofline and CPU-baseline objects, and the
    line's own arithmetic.  Nothing about one record relative to another: the boxes differ by several per cent, and a slower
    box must not turn the CPU suite red (ADVICE / VERDICT r04)."""
    for rnd, p in _driver_lines():
        assert p["unit"] == "env steps/s" and p["higher_is_better"] is True and p["n_gpus"] == 1, rnd
        assert p["value"] > 0 and p["ms_per_step"] > 0, rnd
        assert "roofline" in p and "cpu_baseline" in p, rnd
        r = p["roofline"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6, rnd
        tables = p["config"].get("tables_per_gpu")
        if tables:
            assert 0.9 * tables <= p["value"] * p["ms_per_step"] / 1e3 <= tables, rnd


def test_bench_defaults_match_the_driver_contract():
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    # no flags => one GPU and a K/W that finish within minutes; the flags the driver passes exist
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s"' % flag, src), flag
    assert re.search(r'add_argument\("--gpus", type=int, default=1', src)
    # the oracle is only touched by the cpu_baseline leg
    assert "oracle_lib" in src and src.count("import oracle_lib") <= 2


def test_bench_gpus_n_spawns_its_own_ranks_and_refuses_a_mismatch():
    """`python bench.py --gpus 2` from a bare shell launches two ranks itself (torch.distributed.run, rendezvous on 127.0.0.1);
    a WORLD_SIZE that differs from --gpus is an error, never a silent single-rank run (`--gpus N` was ignored).
    The reference covers all games from one process (arena/game.rs:286-296, one_vs_three.rs:55-60)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "ranks": 2, "gpus": 2, "self_launched": True}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in bad.stderr
