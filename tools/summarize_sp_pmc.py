#!/usr/bin/env python3
"""Fold a tools/pmc_sp.sh output directory (per-launch averages of the separate rocprofv3 --pmc passes over mj_k_sp) into
profiles/<round>_sp_pmc.json, which bench.py reads for `roofline_sp`.

  python tools/summarize_sp_pmc.py gpurun_out/r02_pmc_sp2 r02

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* are quad-cycles summed over waves;
SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = active lanes per VALU instruction; FETCH_SIZE / WRITE_SIZE are KiB at the L2's
memory-side interface, FETCH_SIZE doubled on gfx950 as the guide prescribes (WRITE_SIZE calibrated 1.000x in round 1)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    c = {}
    for f in sorted(glob.glob(os.path.join(src, "p*.txt"))):
        for ln in open(f):
            m = re.match(r"(\w+) ([0-9.e+]+) per launch over (\d+)", ln)
            if m:
                c[m.group(1)] = float(m.group(2))
    states = rows = ms = None
    for f in sorted(glob.glob(os.path.join(src, "p*.log"))):
        for ln in open(f, errors="ignore"):
            if ln.startswith("{") and '"sp_phases"' in ln:
                d = json.loads(ln)
                states = d["sp_phases"]["states_per_step"]
                rows = d["sp_phases"]["rows_per_step"]
                ms = d["kernel_ms_per_step"]["mj_k_sp"]
        if states:
            break
    if not states or "SQ_INSTS_VALU" not in c:
        raise SystemExit("incomplete PMC directory")
    n_simd = 256 * 4
    wave_qc = c["SQ_WAVE_CYCLES"]
    waves = c["SQ_WAVES"]
    import time

    from bench import sp_source_sha16

    out = {
        "source": f"tools/pmc_sp.sh -> {src} (bench.py --steps 4 --warmup 2 under rocprofv3 --pmc, mj_k_sp launches only)",
        "source_sha16": sp_source_sha16(), "measured_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "measured_at_tables": 65536,
        "states_per_launch": states, "rows_per_launch": rows, "kernel_ms_under_profiler": ms,
        "valu_insts_per_launch": c["SQ_INSTS_VALU"], "valu_insts_per_state": c["SQ_INSTS_VALU"] / states,
        "salu_insts_per_state": c.get("SQ_INSTS_SALU", 0) / states, "lds_insts_per_state": c.get("SQ_INSTS_LDS", 0) / states,
        "vmem_rd_insts_per_state": c.get("SQ_INSTS_VMEM_RD", 0) / states,
        # share of the SIMDs' time with a VALU instruction in flight: quad-cycles of VALU activity / (SIMDs x kernel quad-cycles),
        # kernel quad-cycles = SQ_WAVE_CYCLES / resident waves (the persistent waves live for the whole launch)
        "valu_busy": c["SQ_ACTIVE_INST_VALU"] / (n_simd * wave_qc / waves) if "SQ_ACTIVE_INST_VALU" in c else None,
        "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64 if "SQ_THREAD_CYCLES_VALU" in c else None,
        "wave_wait_share": c["SQ_WAIT_ANY"] / wave_qc if "SQ_WAIT_ANY" in c else None,
        "wave_issue_stall_share": c["SQ_WAIT_INST_ANY"] / wave_qc if "SQ_WAIT_INST_ANY" in c else None,
        "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if "SQ_LDS_IDX_ACTIVE" in c else None,
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else None,
        "hbm_fetch_bytes_per_state": c["FETCH_SIZE"] * 1024 * 2 / states if "FETCH_SIZE" in c else None,
        "hbm_write_bytes_per_state": c["WRITE_SIZE"] * 1024 / states if "WRITE_SIZE" in c else None,
        "counters_per_launch": c,
    }
    dst = os.path.join(ROOT, "profiles", f"{rnd}_sp_pmc.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "counters_per_launch"}, indent=1))


if __name__ == "__main__":
    main()
