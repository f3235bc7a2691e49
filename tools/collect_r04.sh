#!/bin/bash
# Round-4 evidence run on the GPU box (through gpurun): everything lands under gpurun_out/r04p/ and is copied to profiles/r04/
# by tools/pack_r04.sh.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
# the PMC passes first: bench.py's roofline.traffic quotes profiles/hbm_traffic.json, which is made from them
bash tools/prof_counters.sh r04p/sad_strip > /dev/null 2>&1
bash tools/prof_counters.sh r04p/sad_strip_cfg4 --config cfg4 --steps 10 > /dev/null 2>&1
mkdir -p profiles/r04
for p in sad_strip sad_strip_cfg4; do python tools/pack_profile.py $O/$p profiles/r04/$p; done
python tools/make_hbm_traffic.py > /dev/null && cp profiles/hbm_traffic.json $O/hbm_traffic.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 > $O/bench_cfg4_strong_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 --pairs 8 > $O/bench_cfg4_strong_8pairs_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --launcher threads > $O/bench_threads_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --launcher threads --config cfg4 > $O/bench_threads_cfg4_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --pipeline > $O/bench_pipeline_n1.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/bench_torchrun_world1.json
# the N-rank run rehearsed on this one GPU: two and four ranks, real HIP steps, gloo collectives through host tensors
python bench.py --gpus 2 --backend gloo --device-map 0,0 --config cfg4 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/rehearsal_2ranks_cfg4_strong.json
python bench.py --gpus 4 --backend gloo --device-map 0,0,0,0 --config cfg4 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/rehearsal_4ranks_cfg4_strong.json
python bench.py --gpus 2 --backend gloo --device-map 0,0 --scaling strong --pairs 64 --ref-mode key --pipeline --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/rehearsal_2ranks_key_pipeline.json
python bench.py --gpus 2 --backend gloo --device-map 0,0 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/rehearsal_2ranks_weak.json
bash tools/cfg3_profile.sh r04p/cfg3_chain 20 3 > /dev/null 2>&1
bash tools/cfg3_profile.sh r04p/cfg3_chain_pm16 20 16 > /dev/null 2>&1
python tools/cfg3_time.py > $O/cfg3_chain/stage_times.json 2>/dev/null
mkdir -p $O/cfg5_stream
python tools/stream_latency.py > $O/cfg5_stream/stream_latency_lsq.json 2>/dev/null
python tools/stream_latency.py --ransac > $O/cfg5_stream/stream_latency_ransac.json 2>/dev/null
python tools/almeida_dense_time.py > $O/almeida_lsq_sizes.txt 2>&1
python tools/ransac_time.py > $O/ransac_time.txt 2>&1
python tools/lk_decode_time.py > $O/lk_decode_time.txt 2>&1
python tools/measure_misc.py > $O/misc.json 2>/dev/null
for m in sync ahead; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1000 $m; done > $O/stream_bench_native.txt 2>&1
for b in 4 16 32; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1024 batch $b; done >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 2048 multi 16 0 >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 2048 multi 16 0 0 >> $O/stream_bench_native.txt 2>&1
./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 2048 multi 16 0 0 0 >> $O/stream_bench_native.txt 2>&1
python - <<PY 2>&1 | grep "lk prof" | tail -1 > $O/lk_phase_table.txt
import sys, os
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.set_option("OFPS_HIP_LK_PROF", 1)
fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
for _ in range(3): ctx.lk_flow(fr[0], fr[1], 3, 4, 3, want_entries=False)
PY
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt
# drop the bulky raw traces, keep the csv summaries
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
du -sh $O
