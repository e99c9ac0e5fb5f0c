#!/bin/bash
# Round-5 first GPU call: the tests the housekeeping commit added / changed, then the default bench line.
OUT=/root/repo/gpurun_out/r05_first; mkdir -p $OUT; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py tests/test_gpu_state.py -m gpu -q --durations=8 -k "staggered or rccl or player_state" > $OUT/gputest.log 2>&1; echo "gputest rc=$?"; tail -15 $OUT/gputest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
