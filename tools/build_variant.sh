#!/bin/bash
# Build an experimental kernel variant next to the default library (git-ignored, travels to the GPU box):
#   tools/build_variant.sh w5 -DSP_WGS=5         -> mortal_amd/libmortal_amd_w5.so   (a tag + extra hipcc flags)
#   tools/build_variant.sh c4 -DSP_NS=32
# A/B it with   MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_w5.so python bench.py ...   or  tools/r03_ab2.sh <out> base base w5
set -e
cd /root/repo
tag=$1
shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
    "$@" -o mortal_amd/libmortal_amd_$tag.so mortal_amd/csrc/mj_capi.hip
ls -la mortal_amd/libmortal_amd_$tag.so
