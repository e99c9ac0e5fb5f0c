#!/bin/bash
# Round 6: BASELINE configs[0] against the REAL library once.  Run it HERE (the container with /root/reference):
#   tools/r06_cfg0_gpu.sh
# It copies the reference's mortal/ directory (156 KB of Python, the unchanged script + engine.py + model.py it imports) into the git-ignored
# scratch directory gpurun_scratch/ so that it travels with ONE gpurun call, runs tests/test_one_vs_three_script.py and
# tests/test_reference_engine.py there with MORTAL_AMD_CFG0_REAL=1 (devices cuda:0, no emulator injection: the unchanged script
# loads libmortal_amd.so), keeps the log under profiles/, and deletes the scratch copy.  Nothing of the reference is committed.
set -e
cd /root/repo
mkdir -p gpurun_scratch
rm -rf gpurun_scratch/reference_mortal
cp -r /root/reference/mortal gpurun_scratch/reference_mortal
rm -rf gpurun_scratch/reference_mortal/__pycache__
trap 'rm -rf /root/repo/gpurun_scratch' EXIT
gpurun --timeout 900 -- 'mkdir -p gpurun_out/r06_cfg0; export MORTAL_REF_DIR=$PWD/gpurun_scratch/reference_mortal MORTAL_AMD_CFG0_REAL=1; (python -m pytest tests/test_one_vs_three_script.py tests/test_reference_engine.py -x -v -s -p no:cacheprovider 2>&1 | grep -v "Warning\|warn" | tail -40) > gpurun_out/r06_cfg0/log.txt; echo "libraries mapped by the script subprocess are the ones of this tree: $(ls -la mortal_amd/libmortal_amd.so | cut -c1-80)" >> gpurun_out/r06_cfg0/log.txt; cat gpurun_out/r06_cfg0/log.txt'
cp gpurun_out/r06_cfg0/log.txt profiles/r06_cfg0_real_library.log
