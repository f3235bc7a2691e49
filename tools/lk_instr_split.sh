#!/bin/bash
# Splits lk_levels_kernel<4>'s VALU / LDS instruction counts into per-step and per-level parts: the same 1080p pair with 1, 2, 3, 4
# Gauss-Newton steps per level (rocprofv3 counters + kernel time).  Per wave-level: count(i) = fixed + G + i * step.
cd $GRAFT_REPO_ROOT
for IT in 1 2 3 4; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sp_t /tmp/sp_w && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_t -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 20 3 $IT > /dev/null 2>&1; timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d /tmp/sp_w -o k -- python $GRAFT_REPO_ROOT/tools/prof_lk.py 10 3 $IT > /dev/null 2>&1)
  IT=$IT python - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/sp_t/**/*kernel_stats.csv", recursive=True)[0]
t = [float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(f)) if "lk_levels" in r["Name"]][0]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/sp_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lk_levels" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1)
print(f"iters {os.environ['IT']}: {t:7.2f} us  waves {w:.0f}  per wave: " + "  ".join(f"{k[3:]} {m[k] / w:8.1f}" for k in sorted(m) if k != "SQ_WAVES"))
PY
done
