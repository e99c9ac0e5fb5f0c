#!/bin/bash
# Run on the GPU box: kernel-trace stats + separate PMC passes for HBM traffic (MI355X_MICROARCH.md §HBM).
# Output under gpurun_out/$1/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-prof}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v4_stats -- python /root/repo/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-matrix --version 4 > $OUT/v4_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v3_stats -- python /root/repo/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-matrix --version 3 > $OUT/v3_stats.log 2>&1
# PMC passes restricted to the encode kernel (--kernel-include-regex): counters slow every profiled launch down
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_write -- python /root/repo/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-matrix --version 4 --tables 16384 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_fetch -- python /root/repo/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-matrix --version 4 --tables 16384 > $OUT/pmc_fetch.log 2>&1
tail -1 $OUT/v4_stats.log | cut -c1-400
tail -1 $OUT/v3_stats.log | cut -c1-400
find $OUT -name "*.csv" | head -20
