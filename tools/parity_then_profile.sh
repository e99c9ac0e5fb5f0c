#!/bin/bash
# Last GPU call of the round: parity of the new build; when green, its rocprofv3 kernel stats + SP phase timers on the
# default bench; when not, the same for mortal_amd/libmortal_amd_prev.so (the last validated build).
cd /root/repo; mkdir -p gpurun_out/fin; export TMPDIR=/tmp
timeout 45 python -m pytest tests/test_gpu_state.py tests/test_gpu_parity.py -m gpu -x -q -k "random_hands or greedy_policy_v4" > gpurun_out/fin/parity.log 2>&1
rc=$?; echo "parity rc=$rc"; tail -3 gpurun_out/fin/parity.log
lib=libmortal_amd.so; [ $rc -ne 0 ] && { lib=libmortal_amd_prev.so; tail -c 3000 gpurun_out/fin/parity.log; }
echo "bench on $lib"
cd /tmp && MJ_SP_PROF=1 MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fin/prof -- \
    python /root/repo/bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 > /root/repo/gpurun_out/fin/bench.json 2> /root/repo/gpurun_out/fin/bench.err
echo "bench rc=$?"; cat /root/repo/gpurun_out/fin/bench.json | cut -c1-400; grep -a "sp prof" /root/repo/gpurun_out/fin/bench.err | tail -2
