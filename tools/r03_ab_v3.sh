#!/bin/bash
# v3 / v4 encode A/B of library variants: tools/r03_ab_v3.sh <outdir-tag> base ep8 ...
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  timeout 120 python bench.py --no-cpu-baseline --no-matrix --version 3 --steps 300 --warmup 20 > $OUT/v3_$tag.json 2> $OUT/v3_$tag.err
  timeout 120 python bench.py --no-cpu-baseline --no-matrix --version 4 --steps 20 --warmup 5 > $OUT/v4_$tag.json 2>> $OUT/v3_$tag.err
  python - <<PY
import json
for f in ("v3", "v4"):
    try:
        d = json.load(open(f"$OUT/{f}_$tag.json")); print("   $lib", f, round(d["value"]), round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, "enc frac", round(d["roofline"]["frac"], 3))
    except Exception as e: print("   $lib", f, "no result", e)
PY
done
