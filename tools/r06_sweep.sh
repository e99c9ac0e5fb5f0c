#!/bin/bash
# Round 6: (wide grid, level-1 threshold) sweep of the small-pool schedule.  tools/r06_sweep.sh <out> "<sizes>" "<grids>" "<min1s>" [extra env]
OUTTAG=$1; SIZES=$2; GRIDS=$3; MINS=$4; EXTRA=$5
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for n in $SIZES; do
  for g in $GRIDS; do
    for m in $MINS; do
      env $EXTRA MJ_SP_WIDE=1 MJ_SP_WIDE_GRID=$g MJ_SP_PROMO_MIN1=$m MJ_SP_PROMO_MIN2=${MIN2:-1000000} timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 40 --warmup 5 --tables $n > $OUT/s_${n}_${g}_${m}.json 2> $OUT/s_${n}_${g}_${m}.err
      python - <<PY
import json
try:
    d = json.load(open("$OUT/s_${n}_${g}_${m}.json"))
    s = d.get("sp_schedule", {})
    print("n $n grid $g min1 $m: sp", round(d["kernel_ms_per_step"]["mj_k_sp"], 3), "ms  promoted/launch", round(s.get("rows_promoted", 0) / max(1, s.get("hybrid_launches", 1)), 1), "swept", s.get("rows_swept"), "gave_up", s.get("wide_gave_up"))
except Exception as e:
    print("n $n grid $g min1 $m: no result", e)
PY
    done
  done
done
