#!/bin/bash
# Round-5 evidence run on the GPU box (through gpurun): everything lands under gpurun_out/r05p/ and is copied to profiles/r05/
# by tools/pack_r05.sh.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
# the PMC passes first: bench.py's roofline.traffic quotes profiles/hbm_traffic.json, which is made from them
bash tools/prof_counters.sh r05p/sad_strip > /dev/null 2>&1
bash tools/prof_counters.sh r05p/sad_strip_cfg4 --config cfg4 --steps 10 > /dev/null 2>&1
mkdir -p profiles/r05
for p in sad_strip sad_strip_cfg4; do python tools/pack_profile.py $O/$p profiles/r05/$p; done
python tools/make_hbm_traffic.py > /dev/null && cp profiles/hbm_traffic.json $O/hbm_traffic.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python tools/perf_gate.py $O/bench_n1.json > $O/perf_gate.txt 2>&1
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 > $O/bench_cfg4_strong_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 --pairs 8 > $O/bench_cfg4_strong_8pairs_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --launcher threads > $O/bench_threads_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --pipeline > $O/bench_pipeline_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --content camera --sad-mode pruned > $O/bench_pruned_camera_n1.json 2>/dev/null
bash tools/cfg3_profile.sh r05p/cfg3_chain 20 3 > /dev/null 2>&1
bash tools/cfg3_profile.sh r05p/cfg3_chain_pm16 20 16 > /dev/null 2>&1
python tools/cfg3_time.py > $O/cfg3_chain/stage_times.json 2>/dev/null
mkdir -p $O/cfg5_stream
python tools/stream_latency.py > $O/cfg5_stream/stream_latency_lsq.json 2>/dev/null
python tools/stream_latency.py --ransac > $O/cfg5_stream/stream_latency_ransac.json 2>/dev/null
python tools/almeida_dense_time.py > $O/almeida_lsq_sizes.txt 2>&1
python tools/ransac_time.py > $O/ransac_time.txt 2>&1
python tools/lk_decode_time.py > $O/lk_decode_time.txt 2>&1
python tools/measure_misc.py > $O/misc.json 2>/dev/null
python tools/sad_geometry_time.py 2>/dev/null | tail -1 > $O/sad_geometry_time.json
for m in sync ahead; do for i in 1 2 3; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1000 $m; done; done > $O/stream_bench_native.txt 2>&1
for b in 4 16 32; do for i in 1 2 3; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1024 batch $b; done; done >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 2048 multi 16 0 >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 2048 multi 16 0 0 >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 3840 2160 256 multi 4 0 0 0 0 0 0 0 0 --block 8 --range 32 >> $O/stream_bench_native.txt 2>&1
# hip_flow (farneback.hip): time, per-dispatch sequence, kernel stats, counters
python tools/farneback_time.py 30 > $O/farneback_time.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/fb_trace && rocprofv3 --kernel-trace --stats --output-format csv -d $O/fb_trace -o k -- python $R/tools/farneback_time.py 6 > /dev/null 2>&1)
python tools/kstats.py $O/fb_trace/k_kernel_stats.csv > $O/farneback_kernel_stats.txt 2>&1
python tools/ktrace_seq.py $O/fb_trace/k_kernel_trace.csv fb_pyr_h > $O/farneback_dispatch_sequence.txt 2>&1
bash tools/fb_counters.sh > $O/farneback_counters.txt 2>&1
./tools/ubench_f64 > $O/ubench_f64.txt 2>&1; ./tools/ubench_dpp > $O/ubench_dpp.txt 2>&1
python tools/accuracy_clips.py --out $O/accuracy_table.txt --json $O/accuracy.json > /dev/null 2> $O/accuracy.err
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt
# drop the bulky raw traces, keep the csv summaries
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/fb_trace $R/gpurun_out/r05x/fb_pmc
du -sh $O
