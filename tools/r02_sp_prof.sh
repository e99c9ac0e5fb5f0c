#!/bin/bash
# One bench run with the mj_k_sp phase / workgroup-lifetime timers on (MJ_SP_PROF=1); prints the [sp prof] lines.
cd /root/repo; mkdir -p gpurun_out/prof
MJ_SP_PROF=1 timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 "$@" > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
grep -a "sp prof" gpurun_out/prof/bench.err | tail -2
python -c "
import json; d=json.load(open('gpurun_out/prof/bench.json')); print(round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'], d.get('sp_phases'))"
