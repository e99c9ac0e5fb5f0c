#!/bin/bash
# v3 bench on the -DMJ_STEP_PROF variant (tools/build_variant.sh stepprof -DMJ_STEP_PROF): where mj_k_step's wave time goes.
cd /root/repo; mkdir -p gpurun_out/stepprof
export MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_${1:-stepprof}.so
MJ_STEP_PROF=1 timeout 200 python bench.py --no-cpu-baseline --no-matrix --version 3 --steps 200 --warmup 20 > gpurun_out/stepprof/bench.json 2> gpurun_out/stepprof/bench.err
grep -a "step prof" gpurun_out/stepprof/bench.err | tail -2
python -c "
import json; d=json.load(open('gpurun_out/stepprof/bench.json')); print(round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
