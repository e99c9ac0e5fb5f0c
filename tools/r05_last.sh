#!/bin/bash
# Round 5, last call: step-path A/B of the deal / kyoku-init change (library tags hd = HEAD, dl = change), the GPU tests that run the step /
# replay kernels with the change as the default library, and the mj_k_sp PMC summary re-stamped (mj_rules.h is part of its source stamp).
cd /root/repo; mkdir -p gpurun_out/r05_last
tools/r05_step_ab.sh r05_last hd dl hd dl
cp mortal_amd/libmortal_amd_dl.so mortal_amd/libmortal_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py tests/test_gpu_arena.py tests/test_dataset.py tests/test_gpu_kats.py -m gpu -x -q -k "not v4 and not full_size" > gpurun_out/r05_last/gputest.log 2>&1; echo "gputest rc=$?"; tail -3 gpurun_out/r05_last/gputest.log
tools/pmc_sp.sh r05_last_pmc_sp 65536 > /dev/null 2>&1; python tools/summarize_sp_pmc.py gpurun_out/r05_last_pmc_sp r05 | tail -2; cp profiles/r05_sp_pmc.json gpurun_out/r05_last/
