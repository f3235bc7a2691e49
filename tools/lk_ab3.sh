#!/bin/bash
# A/B of generated row schedules on the GPU box: each argument is "<generator environment>::<extra hipcc flags>", e.g.
# "LK_GEN_NQ=4 LK_GEN_QBASE=80::-DOFPS_LK_WAVES4=5".  Regenerates lk_rows9.inc, rebuilds, prints the cfg3 leg's parity verdict and
# the rocprofv3 average of lk_levels_kernel; restores the default schedule and build at the end.
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  GENV="${V%%::*}"; FL="${V#*::}"
  env $GENV python tools/gen_lk_rows9.py > /dev/null || { echo "generator failed for '$GENV'"; continue; }
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > /tmp/ab_build.log 2>&1 || { echo "build failed for '$V'"; tail -5 /tmp/ab_build.log; continue; }
  echo "=== gen '$GENV' flags '$FL'"
  python bench_legs.py cfg3_chain 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['cfg3_chain']; print('  leg', {k:(v['lk_ms'],v['chain_ms']) for k,v in d['per_content'].items()}, 'parity', d['parity_check']['ok'])"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_t && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_t -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 20 > /dev/null 2>&1)
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ab_t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lk_levels" in r["Name"]: print(f"  {r['Name'][:40]:40s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
done
python tools/gen_lk_rows9.py > /dev/null; python -m ofps_amd.build --force > /dev/null 2>&1
