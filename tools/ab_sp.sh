#!/bin/bash
# One GPU call: SP parity on the freshly built library, then bench A/B against mortal_amd/libmortal_amd_prev.so
# (the library of the previous commit, copied there by hand; git-ignored).  Output under gpurun_out/ab/.
cd /root/repo; mkdir -p gpurun_out/ab
timeout 100 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q \
    -k "random_hands or greedy_policy_v4 or v4_full_obs" > gpurun_out/ab/parity.log 2>&1
echo "parity rc=$?" | tee -a gpurun_out/ab/parity.log
tail -5 gpurun_out/ab/parity.log
for lib in libmortal_amd_prev.so libmortal_amd.so; do
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
