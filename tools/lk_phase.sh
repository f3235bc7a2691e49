#!/bin/bash
# Level-0 phase table (OFPS_HIP_LK_PROF: per-workgroup clock stamps at the phase boundaries of the first step) for each set of extra
# hipcc flags.  usage (GPU box): lk_phase.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for FL in "$@"; do
  OFPS_HIP_EXTRA_FLAGS="$FL" python -m ofps_amd.build --force > /dev/null 2>&1
  echo "=== '$FL'"
  python - <<PY 2>&1 | grep "lk prof" | tail -1
import sys, os
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.set_option("OFPS_HIP_LK_PROF", 1)
fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
for _ in range(3): ctx.lk_flow(fr[0], fr[1], 3, 4, 3, want_entries=False)
PY
done
python -m ofps_amd.build --force > /dev/null 2>&1
