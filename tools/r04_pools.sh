#!/bin/bash
# --pools K A/B (one library, one box).   tools/r04_pools.sh <outdir-tag> "1 2 4 1"
OUTTAG=$1; cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for k in $2; do
  timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 --pools $k > $OUT/bench_p$k.json 2> $OUT/bench_p$k.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_p$k.json"))
    print("   pools $k:", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, "games/s", round(d["games_per_sec"]))
except Exception as e:
    print("   pools $k: no result", e); print(open("$OUT/bench_p$k.err").read()[-800:])
PY
done
