#!/bin/bash
# rocprofv3 PMC passes of one 1080p Farneback pair (tools/farneback_time.py), summarised per kernel: gpurun_out/r05/fb_pmc/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05x/fb_pmc; rm -rf $O; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tools/farneback_time.py 6"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o k -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o k -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o k -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o k -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05x/fb_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if not n.startswith(("fb_", "void fb_")):
            continue
        g = int(r["Grid_Size"]) if "Grid_Size" in r else 0
        acc[(n, g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, g), c in sorted(acc.items(), key=lambda kv: -kv[0][1])[:8]:
    m = {k: sum(v) / len(v) for k, v in c.items()}
    w = m.get("SQ_WAVES", 1)
    print(f"{n[:28]:<28} grid {g:>9}  waves {w:8.0f}  VALU/wave {m.get('SQ_INSTS_VALU', 0) / w:7.0f}  SALU/wave {m.get('SQ_INSTS_SALU', 0) / w:6.0f}  LDS/wave {m.get('SQ_INSTS_LDS', 0) / w:6.0f}  "
          f"valu_active/busy {m.get('SQ_ACTIVE_INST_VALU', 0) / max(m.get('SQ_BUSY_CYCLES', 1), 1) / 4:5.2f}  lds_active/gui {m.get('SQ_ACTIVE_INST_LDS', 0) / max(m.get('GRBM_GUI_ACTIVE', 1), 1) / 256:5.2f}  "
          f"bank_conf/lds_active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1):5.2f}  FETCH {m.get('FETCH_SIZE', 0) / 1024:7.1f} MB(x2)  WRITE {m.get('WRITE_SIZE', 0) / 1024:7.1f} MB")
PY
