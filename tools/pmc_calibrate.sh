#!/bin/bash
# Calibrate rocprofv3 WRITE_SIZE on a known byte count: torch fill of a 1 GiB f32 buffer (1,073,741,824 B written).
OUT=/root/repo/gpurun_out/${1:-cal}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal -- python -c "
import torch
x=torch.empty(2**28,dtype=torch.float32,device='cuda')
for _ in range(5): x.fill_(1.0)
torch.cuda.synchronize()" > $OUT/cal.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/cal/*/*counter_collection.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='WRITE_SIZE': print(r['Kernel_Name'][:60], r['Grid_Size'], float(r['Counter_Value'])*1024/2**30)
PY
