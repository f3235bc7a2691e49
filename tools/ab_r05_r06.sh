#!/bin/bash
# Same-box A/B of the cfg3 leg: the round-5 tree against this tree, alternating processes.  The round-5 tree is made in the build container first:
#   mkdir -p build/r05tree && git archive 2bdfaf5 | tar -x -C build/r05tree && (cd build/r05tree && python -c "from ofps_amd import build; build.build(); import oracle; oracle.build()")
# (build/ is git-ignored but travels to the GPU box with the snapshot).
# usage (on the GPU box): tools/ab_r05_r06.sh [repeats]   -> gpurun_out/r06/ab_r05_r06.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
N=${1:-3}
pick='import json,sys; d=json.load(sys.stdin)["cfg3_chain"]; p=d["per_content"]; r=d["decoders_read_ahead"]; print(sys.argv[1], "lk pm3", p["pm3"]["lk_ms"], "pm16", p["pm16"]["lk_ms"], "farneback", p["pm3"]["farneback_ms"], "chain", p["pm3"]["chain_ms"], "alm", p["pm3"]["almeida_ms"], "dec hip_lk", r["hip_lk"]["ms_per_frame"], "hip_flow", r["hip_flow"]["ms_per_frame"])'
for i in $(seq $N); do
  (cd $R/build/r05tree && python bench_legs.py cfg3_chain 2>/dev/null | python -c "$pick" r05) 
  (cd $R && python bench_legs.py cfg3_chain 2>/dev/null | python -c "$pick" r06)
done | tee $O/ab_r05_r06.txt
