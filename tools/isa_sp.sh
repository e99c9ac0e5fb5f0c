#!/bin/bash
# Device ISA of the library + a per-function summary of the SP kernel's phase functions (static review of address spaces,
# spills and wait counts; no GPU needed).  Output: gpurun_out/isa/sp.s
set -e
OUT=/root/repo/gpurun_out/isa; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value --cuda-device-only -S \
    -o $OUT/sp.s /root/repo/mortal_amd/csrc/mj_capi.hip 2>&1 | grep -v "argument unused" || true
for f in $(grep -o "^_Z[0-9]*sp_[A-Za-z0-9_]*" $OUT/sp.s | sort -u) _Z7mj_k_sp8SpParams; do
  s=$(grep -n "^$f:" $OUT/sp.s | head -1 | cut -d: -f1)
  [ -z "$s" ] && continue
  awk -v s=$s 'NR>=s' $OUT/sp.s | awk '/^\.Lfunc_end/{exit} {print}' > $OUT/$f.s
  echo "$f lines $(wc -l < $OUT/$f.s) flat $(grep -c 'flat_' $OUT/$f.s) global $(grep -c 'global_' $OUT/$f.s) ds $(grep -c 'ds_' $OUT/$f.s) scratch $(grep -c scratch_ $OUT/$f.s) vgpr $(grep "^\s*\.set \.L$f\.num_vgpr," $OUT/sp.s | head -1 | awk '{print $NF}')"
done
grep -n "amdhsa_kernel _Z7mj_k_sp" -A 40 $OUT/sp.s | grep -i "next_free_vgpr\|private_segment_fixed\|group_segment_fixed"
