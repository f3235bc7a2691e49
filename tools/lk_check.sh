#!/bin/bash
# LK parity tests + cfg3 stage times + per-kernel rocprofv3 averages + the level-0 phase table, for the build in the tree.
# usage (GPU box): lk_check.sh <out-subdir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_reference_vectors_gpu.py tests/test_plugins_gpu.py tests/test_gpu_properties.py -x -q -m gpu -k "lk or flow or golden or decode or 1080p" > $OUT/t.log 2>&1
tail -n 3 $OUT/t.log
python tools/cfg3_time.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: (v['lk_flow_ms'], v['chain_ms']) for k, v in d.items()})" | tee $OUT/times.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 20 > /dev/null 2>&1)
python - <<PY | tee -a $OUT/times.txt
import csv, glob
f = glob.glob("/tmp/tr/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lk_" in r["Name"]:
        print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
python - <<PY 2>&1 | grep "lk prof" | tail -1 | tee -a $OUT/times.txt
import sys, os
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.set_option("OFPS_HIP_LK_PROF", 1)
fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
for _ in range(3): ctx.lk_flow(fr[0], fr[1], 3, 4, 3, want_entries=False)
PY
