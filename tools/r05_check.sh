#!/bin/bash
# Round-5 checkpoint call: the default bench line (workload matrix + CPU baseline) and the rocprofv3 kernel stats of a short v4 run.
TAG=${1:-r05_check}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT/profiles; cd /root/repo
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
cp $OUT/bench.json $OUT/profiles/r05_bench_v4.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v4_stats -- python /root/repo/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-matrix --version 4 > $OUT/v4_stats.log 2>&1
f=$(find $OUT/v4_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/profiles/r05_bench_v4_kernel_stats.csv && head -8 $f | cut -c1-160
rm -rf $OUT/v4_stats
