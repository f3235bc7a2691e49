#!/bin/bash
# A/B of compile-time switches of almeida.hip on the GPU box: one rebuild per flag set, tools/almeida_dense_time.py each.
# usage: almeida_flags_ab.sh <out-subdir under gpurun_out> "<flags>" "<flags>" ...
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for FL in "$@"; do
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > /dev/null 2>&1
  echo "=== '$FL'" | tee -a $OUT/ab.txt
  python tools/almeida_dense_time.py 2>/dev/null | tee -a $OUT/ab.txt
done
python -m ofps_amd.build --force > /dev/null 2>&1
