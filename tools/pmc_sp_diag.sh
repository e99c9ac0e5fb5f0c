#!/bin/bash
# Round 4: where does mj_k_sp's shared bottleneck sit?  (A leaner evaluation made the OTHER phases slower: something shared saturates.)
# Memory-pipeline / instruction-cache / TLB / LDS counters of mj_k_sp only, separate rocprofv3 --pmc passes.
#   tools/pmc_sp_diag.sh <outdir-tag> [tables]
OUT=/root/repo/gpurun_out/${1:-pmc_sp_diag}; TABLES=${2:-65536}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_ATOMIC_WAVEFRONT_sum" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY" \
  "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
  "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); tag=d$i
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex mj_k_sp --output-format csv -d $OUT/$tag -- \
      python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-matrix --version 4 --tables $TABLES > $OUT/$tag.log 2>&1
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
  rm -rf $OUT/$tag
done
