#!/bin/bash
# A/B of libofps_hip.so builds on the GPU box: for each set of extra hipcc flags ("" = the default build) rebuild in place and print
# the cfg3 leg's LK times on both contents with its parity verdict, the rocprofv3 kernel averages and WRITE_SIZE / VALU instructions of
# the LK kernels.  usage: lk_ab2.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for FL in "$@"; do
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > /dev/null 2>&1
  echo "=== '$FL'"
  python bench_legs.py cfg3_chain 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['cfg3_chain']; print('  leg', {k:(v['lk_ms'],v['chain_ms']) for k,v in d['per_content'].items()}, 'parity', d['parity_check']['ok'])"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_t /tmp/ab_w && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_t -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 20 > /dev/null 2>&1; timeout 120 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/ab_w -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 10 > /dev/null 2>&1)
  python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ab_t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lk_" in r["Name"]: print(f"  {r['Name'][:48]:48s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ab_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lk_levels" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  lk_levels:", {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
python -m ofps_amd.build --force > /dev/null 2>&1
