#!/bin/bash
# quick loop for farneback.hip work: parity tests, the 1080p pair time, the per-dispatch sequence (gpurun_out/fbq/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fbq; mkdir -p $O; cd $R
python -m pytest tests/test_farneback_gpu.py -x -q 2>&1 | tail -3 > $O/tests.txt
python tools/farneback_time.py 30 > $O/time.json 2>$O/time.err
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/tr && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o k -- python $R/tools/farneback_time.py 6 > /dev/null 2>&1)
python tools/ktrace_seq.py $O/tr/k_kernel_trace.csv fb_pyr_h > $O/seq.txt 2>&1
python tools/kstats.py $O/tr/k_kernel_stats.csv > $O/kstats.txt 2>&1
rm -rf $O/tr
cat $O/tests.txt $O/time.json; cat $O/seq.txt
