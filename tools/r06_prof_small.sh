#!/bin/bash
# Round 6: MJ_SP_PROF timers (workgroup lifetimes: sum / max) of library tags at small pool sizes.   tools/r06_prof_small.sh <out> "<sizes>" tag ...
OUTTAG=$1; SIZES=$2; shift; shift
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  for n in $SIZES; do
    MJ_SP_PROF=1 timeout 150 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 --tables $n > $OUT/prof_${tag}_$n.json 2> $OUT/prof_${tag}_$n.err
    echo "== $tag $n: $(python -c "import json;d=json.load(open('$OUT/prof_${tag}_$n.json'));print(round(d['kernel_ms_per_step']['mj_k_sp'],3),'ms')")"
    grep -a "sp prof" $OUT/prof_${tag}_$n.err | tail -1 | sed 's/.*workgroup lifetimes/workgroup lifetimes/; s/^\[sp prof\] \(rows [0-9]* \)/\1/' 
    grep -a "sp prof" $OUT/prof_${tag}_$n.err | tail -1 | cut -c1-260
  done
done
