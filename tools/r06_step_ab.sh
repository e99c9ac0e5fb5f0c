#!/bin/bash
# Round 6: the step path (wave-cooperative deal) A/B: obs v3 cycle at 65,536 and 4,096 tables, obs v4 at 4,096.  tools/r06_step_ab.sh <out> tag ...
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  for cfg in "3 65536 100" "3 4096 300" "4 4096 100"; do
    set -- $cfg
    timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps $3 --warmup 10 --version $1 --tables $2 > $OUT/st_${tag}_$1_$2.json 2> $OUT/st_${tag}_$1_$2.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/st_${tag}_$1_$2.json"))
    print("$lib v$1 $2 tables:", round(d["value"]), "steps/s", round(d["ms_per_step"], 4), "ms/cycle", {k: round(v, 4) for k, v in d["kernel_ms_per_step"].items()})
except Exception as e:
    print("$lib v$1 $2: no result", e); print(open("$OUT/st_${tag}_$1_$2.err").read()[-600:])
PY
  done
done
