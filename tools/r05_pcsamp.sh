#!/bin/bash
# Round 5: PC-sampling profile of mj_k_sp (rocprofv3 beta feature; first try stochastic, then host_trap), guarded by short timeouts.
# Output: gpurun_out/$1/pcs_<method>/ (csv) + the log.   tools/r05_pcsamp.sh <tag> [lib]
TAG=${1:-r05_pcs}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
[ -n "$2" ] && export MORTAL_AMD_LIB=/root/repo/mortal_amd/$2
cd /tmp && export TMPDIR=/tmp
for m in stochastic host_trap; do
  unit=cycles; iv=1048576; [ $m = host_trap ] && { unit=time; iv=100; }
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv \
      --output-format csv -d $OUT/pcs_$m -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-matrix > $OUT/pcs_$m.log 2>&1
  rc=$?; echo "pc sampling $m rc=$rc"; tail -3 $OUT/pcs_$m.log | cut -c1-300
  find $OUT/pcs_$m -type f | head; du -sh $OUT/pcs_$m
  [ $rc -eq 0 ] && [ -n "$(find $OUT/pcs_$m -name '*pc_sampling*' | head -1)" ] && break
done
# keep the merge small: compress the sample csv files
find $OUT -name '*.csv' -size +1M -exec gzip -f {} \;
du -sh $OUT
