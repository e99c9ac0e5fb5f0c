#!/bin/bash
# rocprofv3 kernel stats of a short v3 bench (step / snapshot / scan / assign / policy / encode per launch)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/v3stats; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python /root/repo/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-matrix --version 3 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:9]:
    print(r['Name'][:50], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'pct', r['Percentage'])
PY
