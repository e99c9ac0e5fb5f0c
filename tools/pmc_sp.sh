#!/bin/bash
# PMC counters of the SP kernel (instruction mix / busy cycles) on a 16384-table pool.
OUT=/root/repo/gpurun_out/${1:-pmc_sp}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python /root/repo/bench.py --steps 3 --warmup 40 --no-cpu-baseline --version 4 --tables 16384 > $OUT/$tag.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
if not fs: print('no output for $tag'); raise SystemExit
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if 'mj_k_sp' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], 'per launch over', n[k])
PY
done
