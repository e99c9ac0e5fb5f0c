#!/bin/bash
# A/B of SP_OPT variants (tools/build_variant.sh style libraries libmortal_amd_<tag>.so) and of the pool-global step path.
cd /root/repo; mkdir -p gpurun_out/ab3
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  [ -f $MORTAL_AMD_LIB ] || { echo "$lib missing"; continue; }
  ( timeout 150 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q -k "random_hands or greedy_policy_v4 or tsumogiri" ) > gpurun_out/ab3/parity_$tag.log 2>&1
  rc=$?; echo "== $lib parity rc=$rc: $(grep -a 'passed\|failed' gpurun_out/ab3/parity_$tag.log | tail -1)"
  [ $rc -ne 0 ] && { grep -a "Error\|assert\|mismatch" gpurun_out/ab3/parity_$tag.log | head -5; continue; }
  MJ_SP_PROF=1 timeout 120 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 > gpurun_out/ab3/bench_$tag.json 2> gpurun_out/ab3/bench_$tag.err
  timeout 120 python bench.py --no-cpu-baseline --no-matrix --version 3 --steps 200 --warmup 20 > gpurun_out/ab3/bench3_$tag.json 2>> gpurun_out/ab3/bench_$tag.err
  python - <<PY
import json
for f in ("bench", "bench3"):
    try:
        d = json.load(open(f"gpurun_out/ab3/{f}_$tag.json"))
        print("   $lib", f, round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"))
    except Exception as e:
        print("   $lib: no result", e)
PY
done
