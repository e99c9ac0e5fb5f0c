#!/bin/bash
# One GPU call: for each library tag given (""=default): SP parity subset, then the default bench with phase timers.
#   tools/r02_ab.sh "" v1 v2 ...
cd /root/repo; mkdir -p gpurun_out/ab2
for tag in "$@"; do
  lib=libmortal_amd.so; [ -n "$tag" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  ( timeout 120 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q \
      -k "random_hands or greedy_policy_v4" ) > gpurun_out/ab2/parity_$tag.log 2>&1
  rc=$?; echo "== $lib parity rc=$rc: $(grep -a 'passed\|failed' gpurun_out/ab2/parity_$tag.log | tail -1)"
  if [ $rc -ne 0 ]; then grep -a "Error\|assert\|mismatch" gpurun_out/ab2/parity_$tag.log | head -5; continue; fi
  MJ_SP_PROF=1 timeout 120 python bench.py --no-cpu-baseline --no-matrix --steps 10 --warmup 3 > gpurun_out/ab2/bench_$tag.json 2> gpurun_out/ab2/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab2/bench_$tag.json"))
    print("   $lib", round(d["value"]), "steps/s", round(d["ms_per_step"], 2), "ms/cycle", {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"))
except Exception as e:
    print("   $lib: no bench result", e)
PY
done
