#!/bin/bash
# A/B of library variants on the v3 cycle only (step / snapshot / encode): parity subset once per tag, then bench --version 3.
cd /root/repo; mkdir -p gpurun_out/ab_v3
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  ( timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tsumogiri or event_logs" ) > gpurun_out/ab_v3/parity_$tag.log 2>&1
  echo "== $lib parity rc=$?: $(grep -a 'passed\|failed' gpurun_out/ab_v3/parity_$tag.log | tail -1)"
  timeout 120 python bench.py --no-cpu-baseline --no-matrix --version 3 --steps 300 --warmup 20 > gpurun_out/ab_v3/bench3_$tag.json 2> gpurun_out/ab_v3/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_v3/bench3_$tag.json"))
    print("   $lib v3", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
except Exception as e:
    print("   $lib: no result", e)
PY
done
