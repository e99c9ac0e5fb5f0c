#!/bin/bash
# Build an experimental kernel variant next to the default library (git-ignored, travels to the GPU box):
#   tools/build_variant.sh 1     -> mortal_amd/libmortal_amd_v1.so   (hipcc ... -DSP_VARIANT=1)
# A/B it with   MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_v1.so python bench.py ...   or  tools/ab_sp.sh v1
set -e
V=${1:-1}
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
    -DSP_VARIANT=$V -o mortal_amd/libmortal_amd_v$V.so mortal_amd/csrc/mj_capi.hip
ls -la mortal_amd/libmortal_amd_v$V.so
