#!/bin/bash
# hip_flow read-ahead decoder, ms per frame, alternating fresh processes of the trees under build/ (git archive <commit> | tar -x -C build/t_<commit>; build) and this tree
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  for D in $R/build/r05tree $R/build/t_* $R; do
    [ -f $D/ofps_amd/libofps_hip.so ] || continue
    echo "$(basename $D) $(cd $D && python $R/tools/fb_decoder_trace.py 300 2>/dev/null | tail -1)"
  done
done
