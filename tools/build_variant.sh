#!/bin/bash
# Build an experimental kernel variant next to the default library (git-ignored, travels to the GPU box):
#   tools/build_variant.sh 6                     -> mortal_amd/libmortal_amd_v6.so   (-DSP_VARIANT=6, see mj_sp.hip "NEXT")
#   tools/build_variant.sh pg -DMJ_POOL_GLOBAL   -> mortal_amd/libmortal_amd_pg.so   (any tag + extra hipcc flags)
# A/B it with   MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_v6.so python bench.py ...   or  tools/ab_sp.sh v6
set -e
cd /root/repo
if [[ "$1" =~ ^[0-9]+$ ]]; then tag=v$1; flags="-DSP_VARIANT=$1"; else tag=$1; flags=""; fi
shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
    $flags "$@" -o mortal_amd/libmortal_amd_$tag.so mortal_amd/csrc/mj_capi.hip
ls -la mortal_amd/libmortal_amd_$tag.so
