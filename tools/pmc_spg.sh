#!/bin/bash
# PMC passes over ONE kernel family of the per-phase SP pipeline.   tools/pmc_spg.sh <outdir-tag> <kernel regex> [passes "1 2 3 4 5"]
OUT=/root/repo/gpurun_out/${1:-pmc_spg}; RE=${2:-mj_k_spg_eval}; PASSES=${3:-"1 2"}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
  "FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1)); tag=p$i
  echo " $PASSES " | grep -q " $i " || continue
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "$RE" --output-format csv -d $OUT/$tag -- \
      python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matrix --version 4 > $OUT/$tag.log 2>&1
  python3 - <<PY | tee $OUT/$tag.txt
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
if not fs: print('no output for $tag ($set)'); raise SystemExit
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    k=(r['Kernel_Name'][:28], r['Counter_Name'])
    acc[k]+=float(r['Counter_Value']); n[k]+=1
for k in sorted(acc): print(k[0], k[1], acc[k]/n[k], 'per launch over', n[k])
PY
  rm -rf $OUT/$tag
done
