#!/bin/bash
# Round-5 quick A/B (same harness as round 3) on the GPU: the v4 bench (30 timed cycles) for every library tag given; `P:<tag>` also runs the SP parity
# subset first, `T:<tag>` also a run with the MJ_SP_PROF timers.     tools/r05_ab.sh <outdir-tag> base P:x T:y ...
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec#*:}; opt=""; [ "$spec" != "$tag" ] && opt=${spec%%:*}
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  if [ "$opt" = "P" ]; then
    ( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 or refill or 4096" ) > $OUT/parity_$tag.log 2>&1
    rc=$?; echo "== $lib parity rc=$rc: $(grep -a 'passed\|failed' $OUT/parity_$tag.log | tail -1)"
    [ $rc -ne 0 ] && { grep -a "Error\|assert\|mismatch" $OUT/parity_$tag.log | head -8; continue; }
  fi
  if [ "$opt" = "T" ]; then
    MJ_SP_PROF=1 timeout 150 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 > $OUT/benchprof_$tag.json 2> $OUT/benchprof_$tag.err
    grep -a "sp prof" $OUT/benchprof_$tag.err | tail -1 > $OUT/spprof_$tag.txt; cat $OUT/spprof_$tag.txt
  fi
  timeout 150 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json"))
    print("   $lib", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"), "overflow", d.get("sp_phases", {}).get("overflows"))
except Exception as e:
    print("   $lib: no result", e); print(open("$OUT/bench_$tag.err").read()[-600:])
PY
done
