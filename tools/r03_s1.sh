#!/bin/bash
# Round-3 GPU call 1: what the SQ can count on gfx950, the issue rates of the instruction classes mj_k_sp is made of
# (tools/ubench_valu.hip), the baseline phase split, and instruction-TYPE counters of mj_k_sp.  Output: gpurun_out/$1/.
TAG=${1:-r03a}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo
/root/repo/tools/bin/ubench_valu > $OUT/ubench.jsonl 2> $OUT/ubench.err; echo "ubench rc=$? lines $(wc -l < $OUT/ubench.jsonl)"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters_raw.txt 2>&1
grep -o "SQ[A-Z]*_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" $OUT/counters_raw.txt | sort -u > $OUT/counters.txt
echo "counters: $(wc -l < $OUT/counters.txt)"; rm -f $OUT/counters_raw.txt
cd /root/repo
MJ_SP_PROF=1 timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 20 --warmup 5 > $OUT/bench_prof.json 2> $OUT/bench_prof.err
grep -a "sp prof" $OUT/bench_prof.err | tail -3
timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
for f in ("bench_prof", "bench"):
    try:
        d = json.load(open("$OUT/%s.json" % f)); print(f, round(d["value"]), d["ms_per_step"], d["kernel_ms_per_step"], d.get("sp_phases"))
    except Exception as e: print(f, "no result", e)
PY
# instruction-type counters, 7 per pass, only names this rocprofv3 knows
WANT="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_SCRATCH SQ_INSTS_EXP_GDS SQ_INSTS_GDS SQ_WAVES_EQ_64 SQ_WAVES_LT_64 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_ITEMS"
HAVE=""; for c in $WANT; do grep -qx "$c" $OUT/counters.txt && HAVE="$HAVE $c"; done
echo "type counters available: $HAVE" | tee $OUT/type_counters.txt
cd /tmp
set -- $HAVE; i=0
while [ $# -gt 0 ]; do
  grp=""; n=0; while [ $# -gt 0 ] && [ $n -lt 7 ]; do grp="$grp $1"; shift; n=$((n+1)); done
  i=$((i+1)); tag=t$i
  timeout 200 rocprofv3 --pmc $grp --kernel-include-regex mj_k_sp --output-format csv -d $OUT/$tag -- \
      python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-matrix --version 4 > $OUT/$tag.log 2>&1
  python3 - <<PY | tee $OUT/$tag.txt
import csv,glob,collections
fs=glob.glob('$OUT/$tag/*/*counter_collection.csv')
if not fs: print('no output for $tag ($grp)'); raise SystemExit
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if 'mj_k_sp' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], 'per launch over', n[k])
PY
  grep -a '"sp_phases"' $OUT/$tag.log | head -1 | cut -c1-100 > /dev/null
  rm -rf $OUT/$tag
done
du -sh $OUT
