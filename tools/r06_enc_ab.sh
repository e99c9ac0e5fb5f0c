#!/bin/bash
# Round 6: encoder variants, obs v3 and v4 at 65,536 tables.  tools/r06_enc_ab.sh <out> tag ...
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  for v in 3 4; do
    steps=60; [ $v = 4 ] && steps=20
    timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps $steps --warmup 5 --version $v > $OUT/enc_${tag}_v$v.json 2> $OUT/enc_${tag}_v$v.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/enc_${tag}_v$v.json"))
    r = d["roofline"]
    print("$lib v$v:", round(d["ms_per_step"], 3), "ms/cycle  encode", round(d["kernel_ms_per_step"]["mj_k_encode"], 4), "ms  frac of spec", round(r["frac"], 3), " of measured write ceiling", r.get("frac_of_measured_write_ceiling"), "ceiling GB/s", r.get("measured_write_ceiling_GBps") or r.get("measured_write_ceiling"))
except Exception as e:
    print("$lib v$v: no result", e); print(open("$OUT/enc_${tag}_v$v.err").read()[-600:])
PY
  done
done
