#!/bin/bash
# One GPU call: SP / shanten parity of a library, then the default bench on it (and on the default library for reference).
#   tools/ab_sp.sh            parity + bench of mortal_amd/libmortal_amd.so
#   tools/ab_sp.sh v1         the same for mortal_amd/libmortal_amd_v1.so (tools/build_variant.sh 1), then the default bench
# Output under gpurun_out/ab/.
cd /root/repo; mkdir -p gpurun_out/ab
lib=libmortal_amd.so; [ -n "$1" ] && lib=libmortal_amd_$1.so
export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
( timeout 60 python __graft_entry__.py smoke && timeout 80 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q \
    -k "random_hands or greedy_policy_v4 or greedy_policy_v3 or size_independent or tsumogiri" ) > gpurun_out/ab/parity.log 2>&1
rc=$?; echo "parity of $lib rc=$rc" | tee -a gpurun_out/ab/parity.log
grep -a "smoke\|passed\|failed\|Error" gpurun_out/ab/parity.log | tail -6
if [ $rc -ne 0 ]; then tail -c 6000 gpurun_out/ab/parity.log; exit 1; fi   # no bench on a library that is not bit-exact
libs="$lib"; [ "$lib" != libmortal_amd.so ] && libs="$lib libmortal_amd.so"
for l in $libs; do
  MORTAL_AMD_LIB=/root/repo/mortal_amd/$l MJ_SP_PROF=1 timeout 80 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 \
      > gpurun_out/ab/bench_$l.json 2> gpurun_out/ab/bench_$l.err
  echo "$l rc=$?"; grep -a "sp prof" gpurun_out/ab/bench_$l.err | tail -1; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab/bench_$l.json"))
    print("$l", round(d["value"]), "steps/s", round(d["ms_per_step"], 2), "ms/cycle", d["kernel_ms_per_step"])
except Exception as e:
    print("$l: no result", e)
PY
done
