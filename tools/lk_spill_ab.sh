#!/bin/bash
# A/B of whole lk.hip variants on the GPU box (put the candidate files under tools/_lkv/ -- untracked -- before the gpurun call): level-kernel time (rocprofv3 kernel stats) and WRITE_SIZE
# (scratch traffic shows up there: the level-0 kernel's records are 32,400 KB per launch).
set -u
cd $GRAFT_REPO_ROOT
cp ofps_amd/csrc/lk.hip /tmp/lk_keep.hip
for V in tools/_lkv/*.hip; do
  cp $V ofps_amd/csrc/lk.hip
  python -m ofps_amd.build > /dev/null 2>&1
  echo "=== $V"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_t /tmp/ab_w && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_t -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 10 > /dev/null 2>&1; timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ab_w -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 10 > /dev/null 2>&1)
  python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ab_t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lk_level_lds" in r["Name"]: print(f"  {r['Name'][:52]:52s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ab_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lk_level_lds" in r["Kernel_Name"]: acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print("  grid", k[0], k[1], round(sum(v) / len(v)))
PY
done
cp /tmp/lk_keep.hip ofps_amd/csrc/lk.hip; python -m ofps_amd.build > /dev/null 2>&1
