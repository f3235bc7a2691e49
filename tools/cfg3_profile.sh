#!/bin/bash
# rocprofv3 evidence for the cfg3 chain (tools/prof_lk.py: LK flow -> densify 150x84 -> Almeida LSQ on 2.07 M records), to be
# run on the GPU box through gpurun.  Kernel trace and each PMC group are separate runs.
# usage: cfg3_profile.sh <out-subdir under gpurun_out> [iterations] [max_step of the content: 3 (default) or 16]
set -u
TAG=${1:-cfg3_chain}; N=${2:-20}; MS=${3:-3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/prof_lk.py $N $MS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- $CMD > $OUT/trace_run.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -o k -- $CMD > $OUT/pmc_sq_run.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o k -- $CMD > $OUT/pmc_sq2_run.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o k -- $CMD > $OUT/pmc_fetch_run.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o k -- $CMD > $OUT/pmc_write_run.txt 2>&1
python $GRAFT_REPO_ROOT/tools/cfg3_profile_summary.py $OUT $N > $OUT/summary.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
