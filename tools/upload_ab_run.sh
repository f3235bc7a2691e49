#!/bin/bash
# GPU-box half of tools/upload_ab.sh: per variant N fresh processes of the single-frame and the batched read-ahead loops
N=${1:-12}
cd $GRAFT_REPO_ROOT
for v in dma k16 k32 k64 k128 product; do
    d=build/ab/up_$v
    for mode in "1000 ahead" "1000 sync" "1024 batch 16" "1024 batch 4"; do
        vals=""
        for i in $(seq $N); do
            r=$(LD_LIBRARY_PATH=$d $d/ofps_hip_tool stream-bench 1920 1080 $mode 2>/dev/null | tail -1 | sed 's/.*"ms_per_frame": \([0-9.]*\).*/\1/')
            vals="$vals $r"
        done
        echo "$v | $mode |$vals"
    done
done
