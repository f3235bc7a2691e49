#!/bin/bash
# raw SQ counters of one farneback.hip kernel (argv 1: substring of the kernel name), one 1080p pair: gpurun_out/r05x/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05x/fb_pmc_raw; rm -rf $O; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tools/farneback_time.py 4"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/p1 -o k -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p2 -o k -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o k -- $CMD > /dev/null 2>&1
python - "$1" <<'PY'
import csv, glob, os, sys, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05x/fb_pmc_raw"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if sys.argv[1] not in n:
            continue
        acc[(n, int(r.get("Grid_Size", 0)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, g), c in sorted(acc.items(), key=lambda kv: -kv[0][1])[:3]:
    print(n, "grid", g)
    for k, v in sorted(c.items()):
        print(f"   {k:<28} {sum(v) / len(v):16.0f}")
PY
rm -rf $O
