#!/bin/bash
# PMC counters of mj_k_sp only (--kernel-include-regex), separate passes, default bench workload at a reduced table count.
#   tools/pmc_sp.sh <outdir-tag> [tables] [extra bench flags]
# Output: gpurun_out/<tag>/<pass>.txt with per-launch averages; tools/summarize_sp_pmc.py folds them into profiles/.
# PMC_SP_PASSES="1 2" restricts the run to those passes (a quick instruction-count / activity check of a kernel variant).
OUT=/root/repo/gpurun_out/${1:-pmc_sp}; TABLES=${2:-65536}; shift; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
  "FETCH_SIZE TCP_TOTAL_ACCESSES_sum" \
  "WRITE_SIZE TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); tag=p$i
  if [ -n "$PMC_SP_PASSES" ] && ! echo " $PMC_SP_PASSES " | grep -q " $i "; then continue; fi
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex mj_k_sp --output-format csv -d $OUT/$tag -- \
      python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-matrix --version 4 --tables $TABLES "$@" > $OUT/$tag.log 2>&1
  python3 - <<PY | tee $OUT/$tag.txt
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
if not fs: print('no output for $tag ($set)'); raise SystemExit
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if 'mj_k_sp' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], 'per launch over', n[k])
PY
  rm -rf $OUT/$tag   # raw csv is large; the per-launch averages are what we keep
done
