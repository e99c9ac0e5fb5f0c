#!/bin/bash
# Round 5: the env-step path (obs v3: step + snapshot + encode<3>, no SP block) for several library tags, with the kernel stats of each.
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  cd /tmp; export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/st_$tag -- python /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-matrix --version 3 > /root/repo/$OUT/bench_$tag.json 2> /root/repo/$OUT/bench_$tag.err
  cd /root/repo
  python - <<PY
import json, glob, csv
d = json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
f = glob.glob("$OUT/st_$tag/**/*kernel_stats.csv", recursive=True)
ks = {r["Name"][:24]: round(float(r["AverageNs"]) / 1e3, 1) for r in csv.DictReader(open(f[0]))} if f else {}
print("   $lib", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle (under rocprofv3); us per launch:", {k: v for k, v in ks.items() if k.startswith(("mj_k_step", "mj_k_snap", "void mj_k_encode<3>"))})
PY
  rm -rf $OUT/st_$tag
done
