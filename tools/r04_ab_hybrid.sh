#!/bin/bash
# Round-4 first call: measure tools/experiments/sp_hybrid_rows.patch (built as mortal_amd/libmortal_amd_hyb.so) against the default
# library inside ONE call: MJ_SP_HYBRID = 1 | 2 | 4, with and without the second stream.   tools/r04_ab_hybrid.sh <outdir-tag>
OUTTAG=${1:-r04hyb}
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
run() {  # run <label> <lib> [env...]
  label=$1; lib=$2; shift 2
  env MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib "$@" timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 30 --warmup 5 \
      > $OUT/bench_$label.json 2> $OUT/bench_$label.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$label.json"))
    print("   $label", round(d["value"]), "steps/s", round(d["ms_per_step"], 3), "ms/cycle", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()}, d.get("sp_phases", {}).get("share"), "overflow", d.get("sp_phases", {}).get("overflows"))
except Exception as e:
    print("   $label: no result", e); print(open("$OUT/bench_$label.err").read()[-600:])
PY
}
run base0 libmortal_amd.so
run base1 libmortal_amd.so
run hyb_off libmortal_amd_hyb.so
for k in 1 2 4; do
  run hyb$k libmortal_amd_hyb.so MJ_SP_HYBRID=$k
  run hyb${k}s libmortal_amd_hyb.so MJ_SP_HYBRID=$k MJ_SP_HYBRID_STREAMS=1
done
run base2 libmortal_amd.so
( MORTAL_AMD_LIB=/root/repo/mortal_amd/libmortal_amd_hyb.so MJ_SP_HYBRID=2 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 or refill or 4096" ) > $OUT/parity_hyb2.log 2>&1
echo "== hyb2 parity rc=$?: $(grep -a 'passed\|failed' $OUT/parity_hyb2.log | tail -1)"
