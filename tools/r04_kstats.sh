#!/bin/bash
# per-kernel durations of the v4 bench (rocprofv3 --kernel-trace --stats).   tools/r04_kstats.sh <outdir-tag> [extra env assignments]
OUTTAG=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$OUTTAG; mkdir -p $OUT
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matrix > $OUT/bench.json 2> $OUT/bench.err
f=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows:
    if "mj_k_sp" in r["Name"] or "encode" in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
rm -rf $OUT/prof
