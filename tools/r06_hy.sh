#!/bin/bash
# Round 6: the small-pool schedule (promotion to mj_k_sp_wide).  tools/r06_hy.sh <out> "<sizes>" "<env settings A>" "<env settings B>" ...
OUTTAG=$1; SIZES=$2; shift; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
i=0
for envs in "$@"; do
  i=$((i+1))
  for n in $SIZES; do
    steps=60; [ $n -ge 32768 ] && steps=30
    env $envs timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps $steps --warmup 5 --tables $n > $OUT/b${i}_$n.json 2> $OUT/b${i}_$n.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/b${i}_$n.json"))
    print("[$envs] $n:", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, "overflow", d.get("sp_phases", {}).get("overflows"), d.get("sp_schedule"))
except Exception as e:
    print("[$envs] $n: no result", e); print(open("$OUT/b${i}_$n.err").read()[-800:])
PY
  done
done
