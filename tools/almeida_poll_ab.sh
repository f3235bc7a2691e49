#!/bin/bash
# A/B of the cluster solver's exchange on the GPU box: extra hipcc flags per variant (-DOFPS_ALMEIDA_POLL=<polls in flight
# per gathering wave>, -DOFPS_ALMEIDA_GRAN_STRIDE=<granules between cross-XCD granules>), each timed with the default
# 1024-thread workgroups and with OFPS_HIP_ALMEIDA_BLOCK=256.
# usage: almeida_poll_ab.sh <out-subdir under gpurun_out> "<flags>" "<flags>" ...
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for FL in "$@"; do
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > $OUT/build_$i.log 2>&1
  for B in ${BLOCKS:-0 256}; do
    echo "=== '$FL' block ${B}" | tee -a $OUT/ab.txt
    if [ $B = 0 ]; then python tools/almeida_dense_time.py 2>/dev/null | tee -a $OUT/ab.txt
    else OFPS_HIP_ALMEIDA_BLOCK=$B python tools/almeida_dense_time.py 2>/dev/null | head -3 | tee -a $OUT/ab.txt; fi
    OFPS_HIP_ALMEIDA_BLOCK=$B python tools/almeida_prof.py 2>&1 >/dev/null | grep -A1 "n=8040 ept=1\|n=2073600" | tee -a $OUT/ab.txt
  done
  i=$((i+1))
done
python -m ofps_amd.build --force > /dev/null 2>&1
