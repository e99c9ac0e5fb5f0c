#!/bin/bash
# Round 6: the encoder against the number of resident workgroups per CU (LDS padding).  tools/r06_enc_occ.sh <out> "<pads>"
OUTTAG=$1; PADS=$2
cd /root/repo; OUT=gpurun_out/$OUTTAG; mkdir -p $OUT
for pad in $PADS; do
  for v in 3 4; do
    steps=60; [ $v = 4 ] && steps=20
    MJ_ENC_LDS_PAD=$pad timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps $steps --warmup 5 --version $v > $OUT/occ_${pad}_v$v.json 2> $OUT/occ_${pad}_v$v.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/occ_${pad}_v$v.json"))
    r = d["roofline"]
    print("pad $pad v$v:", round(d["ms_per_step"], 3), "ms/cycle  encode", round(d["kernel_ms_per_step"]["mj_k_encode"], 4), "ms  of measured write ceiling", round(r.get("frac_of_measured_write_ceiling"), 3))
except Exception as e:
    print("pad $pad v$v: no result", e); print(open("$OUT/occ_${pad}_v$v.err").read()[-600:])
PY
  done
done
