#!/bin/bash
# Round-3 evidence run on the GPU box (through gpurun): everything lands under gpurun_out/r03p/ and is copied to profiles/r03/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
# the PMC passes first: bench.py's roofline.traffic quotes profiles/hbm_traffic.json, which is made from them
bash tools/prof_counters.sh r03p/sad_strip > /dev/null 2>&1
bash tools/prof_counters.sh r03p/sad_strip_cfg4 --config cfg4 --steps 10 > /dev/null 2>&1
for p in sad_strip sad_strip_cfg4; do python tools/pack_profile.py $O/$p profiles/r03/$p; done
python tools/make_hbm_traffic.py > /dev/null && cp profiles/hbm_traffic.json $O/hbm_traffic.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 > $O/bench_cfg4_strong_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --config cfg4 --pairs 8 > $O/bench_cfg4_strong_8pairs_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --launcher threads > $O/bench_threads_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --launcher threads --config cfg4 > $O/bench_threads_cfg4_n1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --pipeline > $O/bench_pipeline_n1.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-end-to-end --no-legs 2>/dev/null | tail -n 1 > $O/bench_torchrun_world1.json
bash tools/cfg3_profile.sh r03p/cfg3_chain 20 > /dev/null 2>&1
python tools/cfg3_time.py > $O/cfg3_chain/stage_times.json 2>/dev/null
mkdir -p $O/cfg5_stream
python tools/stream_latency.py > $O/cfg5_stream/stream_latency_lsq.json 2>/dev/null
python tools/stream_latency.py --ransac > $O/cfg5_stream/stream_latency_ransac.json 2>/dev/null
python tools/almeida_dense_time.py > $O/almeida_lsq_sizes.txt 2>&1
python tools/almeida_prof.py > /dev/null 2> $O/almeida_phase_table.txt
python tools/ransac_time.py > $O/ransac_time.txt 2>&1
python tools/lk_decode_time.py > $O/lk_decode_time.txt 2>&1
python tools/measure_misc.py > $O/misc.json 2>/dev/null
./tools/ubench_lds > $O/ubench_lds.txt 2>&1
./tools/ubench_valu > $O/ubench_valu.txt 2>&1
timeout 120 ./tools/ubench_allgather > $O/ubench_allgather.txt 2>&1
python tools/almeida_block_ab.py > $O/almeida_block_ab.txt 2>/dev/null
for m in sync ahead; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1000 $m; done > $O/stream_bench_native.txt 2>&1
for b in 4 8 16 32; do ./ofps_amd/host/ofps_hip_tool stream-bench 1920 1080 1024 batch $b; done >> $O/stream_bench_native.txt 2>&1
python - <<PY 2>&1 | grep "lk prof" | tail -1 > $O/lk_phase_table.txt
import sys, os
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.set_option("OFPS_HIP_LK_PROF", 1)
fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
for _ in range(3): ctx.lk_flow(fr[0], fr[1], 3, 4, 3, want_entries=False)
PY
# drop the bulky raw traces, keep the csv summaries
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
du -sh $O
