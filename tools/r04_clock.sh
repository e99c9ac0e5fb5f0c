#!/bin/bash
# Does the kernel run at the clock the roofline assumes?  MJ_SP_PROF=1 prints s_memtime cycles / wall_clock64 ticks over the workgroup lifetimes
# of mj_k_sp.  tools/r04_clock.sh <outdir-tag> <lib tags...>
OUTTAG=$1; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib MJ_SP_PROF=1 timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 > $OUT/clk_$tag.json 2> $OUT/clk_$tag.err
  echo "== $lib: $(grep -a -o 'shader clock.*' $OUT/clk_$tag.err | tail -1)"
  python - <<PY
import json
d = json.load(open("$OUT/clk_$tag.json")); print("   ", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
PY
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -a -i "sclk\|power\|mclk" | head -8
