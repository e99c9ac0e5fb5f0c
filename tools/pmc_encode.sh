#!/bin/bash
# PMC WRITE_SIZE of the encode kernel on a 16384-table pool (SP skipped to isolate it)
OUT=/root/repo/gpurun_out/${1:-pmc}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python /root/repo/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-matrix --version 4 --tables 16384 > $OUT/pmc_write.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/pmc_write/*/*counter_collection.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mj_k_encode' in r['Kernel_Name']]
per=[float(r['Counter_Value'])*1024/(int(r['Grid_Size'])/256) for r in rows]
print('WRITE_SIZE bytes/row', sum(per)/len(per))
PY
cd /root/repo; python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-matrix --version 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved'])"
