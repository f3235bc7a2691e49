#!/bin/bash
# kernel stats of the hip_flow read-ahead decoder stream (tools/lk_decode_time.py's last loop): gpurun_out/fbq/stream_kstats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fbq; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/trs && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trs -o k -- python $R/tools/lk_decode_time.py > /dev/null 2>&1)
python $R/tools/kstats.py $O/trs/k_kernel_stats.csv > $O/stream_kstats.txt 2>&1
rm -rf $O/trs
cat $O/stream_kstats.txt
