#!/bin/bash
# A/B of environment-switched variants of ONE library inside one call.   tools/r04_ab_env.sh <outdir-tag> [P] "label:ENV=V ENV2=V2" ...
# "P" as the first spec runs the SP parity subset (default environment) first.
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
if [ "$1" = "P" ]; then
  shift
  ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 or refill or 4096 or staggered" ) > $OUT/parity.log 2>&1
  rc=$?; echo "== parity rc=$rc: $(grep -a 'passed\|failed' $OUT/parity.log | tail -1)"
  [ $rc -ne 0 ] && grep -a "Error\|assert\|mismatch" $OUT/parity.log | head -12
fi
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 > $OUT/bench_$label.json 2> $OUT/bench_$label.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$label.json"))
    print("   $label", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"), "states", d.get("sp_phases", {}).get("states_per_step"), "overflow", d.get("sp_phases", {}).get("overflows"))
except Exception as e:
    print("   $label: no result", e); print(open("$OUT/bench_$label.err").read()[-800:])
PY
done
