#!/bin/bash
# A/B of libofps_hip.so builds on the GPU box: for each set of extra hipcc flags, rebuild in place and print the cfg3 stage
# times (tools/cfg3_time.py) plus the rocprofv3 kernel-time table of the LK kernels.
# usage: lk_ab.sh <out-subdir under gpurun_out> "<flags A>" "<flags B>" ...     ("" = the default build)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
i=0
for FL in "$@"; do
  cd $GRAFT_REPO_ROOT
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > $OUT/build_$i.log 2>&1
  echo "=== variant $i: '$FL'" | tee -a $OUT/ab.txt
  python tools/cfg3_time.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: (v['lk_flow_ms'], v['chain_ms']) for k, v in d.items()})" | tee -a $OUT/ab.txt
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$i -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 20 > /dev/null 2>&1)
  python - <<PY | tee -a $OUT/ab.txt
import csv, glob
f = glob.glob("$OUT/trace_$i/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lk_" in r["Name"]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT && python -m ofps_amd.build --force > /dev/null 2>&1
