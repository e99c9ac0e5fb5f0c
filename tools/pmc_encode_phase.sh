#!/bin/bash
# Instruction-mix PMC counters of the encode kernel early (--preroll 0) vs late (--preroll 70) in a kyoku.
OUT=/root/repo/gpurun_out/${1:-pmc_enc}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pr in 0 70; do
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  tag=pr${pr}_$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python /root/repo/bench.py --steps 4 --warmup 2 --preroll $pr --no-cpu-baseline --no-matrix --version 3 --tables 16384 > $OUT/$tag.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'mj_k_encode' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('preroll $pr', {k: round(sum(v[-4:])/4/1e6,2) for k,v in acc.items()}, 'M per launch (last 4 launches)')
PY
done; done
