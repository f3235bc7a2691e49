#!/bin/bash
# Round-6 evidence run on the GPU box (through gpurun): everything lands under gpurun_out/r06p/ and is copied to profiles/r06/
# by tools/pack_r06.sh.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O
cd $R
# the PMC passes first: bench.py's roofline.traffic quotes profiles/hbm_traffic.json, which is made from them
bash tools/prof_counters.sh r06p/sad_strip > /dev/null 2>&1
bash tools/prof_counters.sh r06p/sad_strip_cfg4 --config cfg4 --steps 10 > /dev/null 2>&1
mkdir -p profiles/r06
for p in sad_strip sad_strip_cfg4; do python tools/pack_profile.py $O/$p profiles/r06/$p; done
python tools/make_hbm_traffic.py > /dev/null && cp profiles/hbm_traffic.json $O/hbm_traffic.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python tools/perf_gate.py $O/bench_n1.json > $O/perf_gate.txt 2>&1
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 > $O/bench_cfg4_strong_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --launcher threads > $O/bench_threads_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --pipeline > $O/bench_pipeline_n1.json 2>/dev/null
bash tools/cfg3_profile.sh r06p/cfg3_chain 20 3 > /dev/null 2>&1
bash tools/cfg3_profile.sh r06p/cfg3_chain_pm16 20 16 > /dev/null 2>&1
python tools/cfg3_time.py > $O/cfg3_chain/stage_times.json 2>/dev/null
python tools/lk_decode_time.py > $O/lk_decode_time.txt 2>&1
python tools/reduced_decoder_time.py > $O/reduced_decoder.txt 2>&1
python tools/measure_misc.py > $O/misc.json 2>/dev/null
python tools/almeida_prof.py 2>&1 | grep "n=2073600" > $O/almeida_prof_2m.txt
# the Almeida solver after the round's last changes: packed record pairs, one-XCD small clusters, short update chain, RANSAC hypotheses
python tools/almeida_dense_time.py > $O/almeida_dense_time.txt 2>&1
python tools/almeida_one_xcd_ab.py 2>&1 | grep -v "per workgroup" > $O/almeida_one_xcd_ab.txt
python tools/almeida_threshold_ab.py > $O/almeida_threshold_ab.txt 2>&1
python tools/almeida_cu_mask_probe.py > $O/almeida_cu_mask_probe.txt 2>&1
python tools/ransac_time.py > $O/ransac_time.txt 2>&1
# hip_flow: time, per-dispatch sequence, kernel stats
python tools/farneback_time.py 30 > $O/farneback_time.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/fb_trace && rocprofv3 --kernel-trace --stats --output-format csv -d $O/fb_trace -o k -- python $R/tools/farneback_time.py 6 > /dev/null 2>&1)
python tools/kstats.py $O/fb_trace/k_kernel_stats.csv > $O/farneback_kernel_stats.txt 2>&1
python tools/ktrace_seq.py $O/fb_trace/k_kernel_trace.csv fb_pyr_h > $O/farneback_dispatch_sequence.txt 2>&1
# the reduced decoder's kernels (front-end, mask, flow on 150 x 84, compaction): rocprofv3 kernel stats of its read-ahead loop
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/red_trace && rocprofv3 --kernel-trace --stats --output-format csv -d $O/red_trace -o k -- python $R/tools/reduced_decoder_time.py 300 > /dev/null 2>&1)
python tools/kstats.py $O/red_trace/k_kernel_stats.csv > $O/reduced_decoder_kernel_stats.txt 2>&1
# robustness: the C ABI fuzz incl. frame formats and the reduced mode, the stream soaks, the forward-progress A/B against the round-5 tree
for seed in 1 2 3 4; do python tools/api_fuzz.py 5000 $seed 2>&1 | tail -1; done > $O/api_fuzz.txt
python tools/flow_soak.py 2>&1 | tail -2 > $O/flow_soak.txt
[ -d build/r05tree ] && bash tools/ab_r05_r06.sh 3 > /dev/null 2>&1 && cp $R/gpurun_out/r06/ab_r05_r06.txt $O/ab_r05_r06.txt
python tools/accuracy_clips.py --out $O/accuracy_table.txt --json $O/accuracy.json > /dev/null 2> $O/accuracy.err
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt
cp gpurun_out/perf_gate.txt $O/perf_gate_test_run.txt 2>/dev/null; cp gpurun_out/perf_gate_line.json $O/bench_n1_gate_run.json 2>/dev/null
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/fb_trace $O/red_trace
du -sh $O
