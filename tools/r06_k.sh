cd /root/repo
MIN2=350 tools/r06_sweep.sh r06_k "4096 8192" "64" "1200"
MIN2=450 tools/r06_sweep.sh r06_k "4096 8192" "64" "1200"
MIN2=1000000 tools/r06_sweep.sh r06_k "4096 8192" "64" "1200"
MIN2=450 tools/r06_sweep.sh r06_k "16384" "32" "1600"
MIN2=1000000 tools/r06_sweep.sh r06_k "16384" "32" "1600"
export MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_w512.so
MIN2=1000000 tools/r06_sweep.sh r06_k5 "4096 8192" "128 192" "900 1200"
MIN2=1000000 tools/r06_sweep.sh r06_k5 "16384" "64 128" "1200 1600"
