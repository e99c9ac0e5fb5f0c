#!/bin/bash
# Round-2 GPU call: the whole -m gpu suite, the default bench line (workload matrix + cpu baseline), rocprofv3 kernel stats of the
# same command, and the PMC passes behind roofline.traffic (encode) and roofline_sp (mj_k_sp).  Output: gpurun_out/$1/.
TAG=${1:-r02}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo
timeout 1300 python -m pytest tests -m gpu -q --durations=15 > $OUT/gputest.log 2>&1; echo "gputest rc=$?"; tail -4 $OUT/gputest.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v4_stats -- python /root/repo/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-matrix --version 4 > $OUT/v4_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v3_stats -- python /root/repo/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-matrix --version 3 > $OUT/v3_stats.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_write -- python /root/repo/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-matrix --version 4 --tables 16384 > $OUT/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_fetch -- python /root/repo/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-matrix --version 4 --tables 16384 > $OUT/pmc_fetch.log 2>&1
cd /root/repo; tools/pmc_sp.sh ${TAG}_pmc_sp 65536 > $OUT/pmc_sp.log 2>&1; tail -3 $OUT/pmc_sp.log
find $OUT -name "*kernel_stats.csv" | head; du -sh $OUT
