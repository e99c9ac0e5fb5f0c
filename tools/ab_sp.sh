#!/bin/bash
# One GPU call: SP / shanten parity on the freshly built library, then the default bench (and, when
# mortal_amd/libmortal_amd_prev.so exists — a hand-made copy of an earlier build, git-ignored — the same bench on it).
# Output under gpurun_out/ab/.   usage: tools/ab_sp.sh [prev]
cd /root/repo; mkdir -p gpurun_out/ab
( timeout 60 python __graft_entry__.py smoke && timeout 80 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q \
    -k "random_hands or greedy_policy_v4" ) > gpurun_out/ab/parity.log 2>&1
rc=$?; echo "parity rc=$rc" | tee -a gpurun_out/ab/parity.log
grep -a "smoke\|passed\|failed\|Error" gpurun_out/ab/parity.log | tail -6
if [ $rc -ne 0 ]; then tail -c 6000 gpurun_out/ab/parity.log; exit 1; fi   # no bench on a library that is not bit-exact
libs="libmortal_amd.so"; [ "$1" = prev ] && libs="libmortal_amd_prev.so libmortal_amd.so"
for lib in $libs; do
  [ -f mortal_amd/$lib ] || continue
  MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib timeout 80 python bench.py --no-cpu-baseline --steps 20 --warmup 5 \
      > gpurun_out/ab/bench_$lib.json 2> gpurun_out/ab/bench_$lib.err
  echo "$lib rc=$?"; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab/bench_$lib.json"))
    print("$lib", round(d["value"]), "steps/s", round(d["ms_per_step"], 2), "ms/cycle", d["kernel_ms_per_step"])
except Exception as e:
    print("$lib: no result", e)
PY
done
