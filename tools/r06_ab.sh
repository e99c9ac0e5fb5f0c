#!/bin/bash
# Round-6 A/B harness on the GPU: for every library tag (interleave them: `base x base x`) the v4 bench at every pool size of $SIZES
# (default "65536"; e.g. SIZES="65536 4096 8192 16384").  `P:<tag>` runs the SP parity subset first.   tools/r06_ab.sh <outdir-tag> base P:x ...
OUTTAG=$1; shift
SIZES=${SIZES:-65536}
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec#*:}; opt=""; [ "$spec" != "$tag" ] && opt=${spec%%:*}
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  if [ "$opt" = "P" ]; then
    ( timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 or refill or 4096" ) > $OUT/parity_$tag.log 2>&1
    rc=$?; echo "== $lib parity rc=$rc: $(grep -a 'passed\|failed' $OUT/parity_$tag.log | tail -1)"
    [ $rc -ne 0 ] && { grep -a "Error\|assert\|mismatch" $OUT/parity_$tag.log | head -8; continue; }
  fi
  for n in $SIZES; do
    steps=30; [ $n -lt 65536 ] && steps=60
    timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps $steps --warmup 5 --tables $n ${BENCH_EXTRA} > $OUT/bench_${tag}_$n.json 2> $OUT/bench_${tag}_$n.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${tag}_$n.json"))
    print("   $lib $n:", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, "overflow", d.get("sp_phases", {}).get("overflows"))
except Exception as e:
    print("   $lib $n: no result", e); print(open("$OUT/bench_${tag}_$n.err").read()[-600:])
PY
  done
done
