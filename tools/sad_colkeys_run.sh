#!/bin/bash
# GPU-box half of the argmin A/B (tools/sad_colkeys_ab.sh builds the three libraries): per variant the SAD parity tests, the cfg4 leg
# (launch_ms, SAD-unit fraction) three times, the headline line twice, and SQ counters of the cfg4 leg.  Output: gpurun_out/r05/sad_colkeys/
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/sad_colkeys
mkdir -p $OUT
cp ofps_amd/libofps_hip.so /tmp/libofps_hip.product.so
for v in 0 1 2; do
    cp build/ab/colkeys$v/libofps_hip.so ofps_amd/libofps_hip.so
    python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_properties.py -m gpu -q -k "sad or golden or SAD" 2>&1 | tail -2 > $OUT/parity_$v.txt
    for rep in 1 2 3; do python bench_legs.py cfg4 2>/dev/null | tail -1 >> $OUT/cfg4_$v.jsonl; done
    for rep in 1 2; do python bench.py --no-legs --no-end-to-end --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/cfg2_$v.jsonl; done
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
        --output-format csv -d $OUT/pmc1_$v -o k -- python $GRAFT_REPO_ROOT/bench_legs.py cfg4 > $OUT/pmc1_$v.txt 2>&1
     rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
        --output-format csv -d $OUT/pmc2_$v -o k -- python $GRAFT_REPO_ROOT/bench_legs.py cfg4 > $OUT/pmc2_$v.txt 2>&1)
done
cp /tmp/libofps_hip.product.so ofps_amd/libofps_hip.so
python - <<'PY'
import csv, glob, json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/sad_colkeys"
for v in (0, 1, 2):
    c4 = [json.loads(l)["cfg4"] for l in open(f"{out}/cfg4_{v}.jsonl")]
    c2 = [json.loads(l) for l in open(f"{out}/cfg2_{v}.jsonl")]
    row = {"variant": v, "parity": open(f"{out}/parity_{v}.txt").read().strip().splitlines()[-1],
           "cfg4_launch_ms": [x["launch_ms"] for x in c4], "cfg4_frac": [x["valu"]["frac"] for x in c4], "cfg4_parity": [x["parity_check"]["ok"] for x in c4],
           "cfg2_Mvec": [x["value"] for x in c2], "cfg2_frac": [x["roofline"]["valu"]["frac"] for x in c2], "cfg2_parity": [x["parity_check"]["ok"] for x in c2]}
    for grp in ("pmc1", "pmc2"):
        acc = {}
        for f in glob.glob(f"{out}/{grp}_{v}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "sad_strip_kernel" in r.get("Kernel_Name", ""):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        row[grp] = {k: round(sum(vv) / len(vv)) for k, vv in acc.items()}
    print(json.dumps(row))
PY
