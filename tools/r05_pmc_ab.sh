#!/bin/bash
# Round 5: PMC passes 1-3 of tools/pmc_sp.sh (instruction counts, activity, LDS conflicts) for several library tags.
#   tools/r05_pmc_ab.sh <outtag> <tag> ...     (tag "base" = libmortal_amd.so); PMC_SP_PASSES overrides the pass list
OUTTAG=$1; shift
export PMC_SP_PASSES="${PMC_SP_PASSES:-1 2 3}"
for tag in "$@"; do
  lib=libmortal_amd.so; [ "$tag" != "base" ] && lib=libmortal_amd_$tag.so
  export MORTAL_AMD_LIB=/root/repo/mortal_amd/$lib
  echo "== $lib"
  /root/repo/tools/pmc_sp.sh ${OUTTAG}_$tag 65536 2>&1 | grep -v "^$"
done
