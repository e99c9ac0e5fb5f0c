#!/bin/bash
# Round 6: what limits mj_k_encode<4>?  Separate rocprofv3 --pmc passes over the encoder only (bench workload, 65,536 tables), per-launch averages.
#   tools/pmc_encode_diag.sh <outdir-tag>        -> gpurun_out/<tag>/p*.txt, counters.txt (the write-side counter names this GPU offers)
OUT=/root/repo/gpurun_out/${1:-pmc_enc_diag}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "\(TCC_EA[A-Z0-9_]*\|TCC_[A-Z_]*WR[A-Z0-9_]*\|TCP_[A-Z_]*WR[A-Z0-9_]*\|TCC_[A-Z_]*STALL[A-Z0-9_]*\|TCP_[A-Z_]*STALL[A-Z0-9_]*\)" | sort -u > $OUT/counters.txt
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum" \
  "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_WRITE_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "WRITE_SIZE FETCH_SIZE"; do
  i=$((i+1)); tag=p$i
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex mj_k_encode --output-format csv -d $OUT/$tag -- \
      python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-matrix --version 4 > $OUT/$tag.log 2>&1
  python3 - <<PY | tee $OUT/$tag.txt
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
if not fs: print('no output for $tag ($set)'); raise SystemExit
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if 'mj_k_encode' in r['Kernel_Name'] and int(r['Grid_Size']) > 256 * 60000:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], 'per launch over', n[k])
PY
  rm -rf $OUT/$tag
done
