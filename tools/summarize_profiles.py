#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the tracked files under profiles/.

  python tools/summarize_profiles.py gpurun_out/<tag> <round>     e.g. gpurun_out/r01b r01

Writes
  profiles/<round>_bench_v{3,4}_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (as produced)
  profiles/<round>_pmc_{write,fetch}_size.csv      per-dispatch counter rows of the mj_* kernels
  profiles/pmc_encode.json                         bytes per decision of mj_k_encode from the PMC passes; bench.py reads it
                                                   for `roofline.traffic` (the PMC passes cannot run inside the timed bench)
Counter handling follows MI355X_MICROARCH.md "HBM": WRITE_SIZE/FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read, so it is doubled; WRITE_SIZE was calibrated on a known torch fill (tools/pmc_calibrate.sh,
1.000x) and is used as is.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[0] if g else None


def pmc_rows(d, name):
    f = first(os.path.join(d, "**", "*counter_collection.csv"))
    if not f:
        return None, []
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("mj_k_") or "mj_k_" in r["Kernel_Name"]]
    return f, [r for r in rows if r["Counter_Name"] == name]


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    tables = int(sys.argv[3]) if len(sys.argv) > 3 else 65536  # table count of the PMC passes (tools/r03_full.sh: the bench's own)
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    for v in (3, 4):
        f = first(os.path.join(src, f"v{v}_stats", "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(prof, f"{rnd}_bench_v{v}_kernel_stats.csv"))
    import time

    out = {"source": f"{src} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes over mj_k_encode)", "tables": tables, "measured_at_tables": tables,
           "obs_version": 4, "measured_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "round": rnd}
    for key, cname, scale in (("write", "WRITE_SIZE", 1.0), ("fetch", "FETCH_SIZE", 2.0)):
        f, rows = pmc_rows(os.path.join(src, f"pmc_{key}"), cname)
        if not rows:
            continue
        with open(os.path.join(prof, f"{rnd}_pmc_{key}_size.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value_KiB"])
            for r in rows:
                w.writerow([r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
        enc = [r for r in rows if "mj_k_encode<4>" in r["Kernel_Name"]]  # the timed v4 launches, not the v3 pre-roll
        per = [float(r["Counter_Value"]) * 1024 * scale / (int(r["Grid_Size"]) / int(r["Workgroup_Size"])) for r in enc]
        out[f"{key}_bytes_per_decision"] = sum(per) / len(per)
        out[f"{key}_dispatches"] = len(per)
        out[f"{key}_scale_applied"] = scale
    with open(os.path.join(prof, "pmc_encode.json"), "w") as o:
        json.dump(out, o, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
