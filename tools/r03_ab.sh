#!/bin/bash
# Round-3 A/B on the GPU: SP parity subset, then the v4 bench with the MJ_SP_PROF phase timers and without, for every library
# tag given (base = mortal_amd/libmortal_amd.so, <tag> = mortal_amd/libmortal_amd_<tag>.so built by tools/build_variant.sh).
#   tools/r03_ab.sh <outdir-tag> base [tag ...]
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  ( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 or refill or 4096" ) > $OUT/parity_$tag.log 2>&1
  rc=$?; echo "== $lib parity rc=$rc: $(grep -a 'passed\|failed' $OUT/parity_$tag.log | tail -1)"
  [ $rc -ne 0 ] && { grep -a "Error\|assert\|mismatch" $OUT/parity_$tag.log | head -8; continue; }
  MJ_SP_PROF=1 timeout 150 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 > $OUT/benchprof_$tag.json 2> $OUT/benchprof_$tag.err
  grep -a "sp prof" $OUT/benchprof_$tag.err | tail -1 > $OUT/spprof_$tag.txt
  timeout 150 python bench.py --no-cpu-baseline --no-matrix --steps 40 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
for f in ("benchprof", "bench"):
    try:
        d = json.load(open(f"$OUT/{f}_$tag.json"))
        print("   $lib", f, round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"), round(d.get("sp_phases", {}).get("states_per_step", 0)))
    except Exception as e:
        print("   $lib: no result", e)
PY
  cat $OUT/spprof_$tag.txt
done
