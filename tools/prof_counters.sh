#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box (run through gpurun).
# Kernel trace/stats and every PMC group are separate runs (gpurun refuses trace+pmc combinations).
# usage: prof_counters.sh <out-subdir> [bench args...]
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-legs $*"   # same steps/warm-up as the default bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- $CMD > $OUT/trace_run.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o k -- $CMD > $OUT/pmc_sq1_run.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o k -- $CMD > $OUT/pmc_sq2_run.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o k -- $CMD > $OUT/pmc_fetch_run.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o k -- $CMD > $OUT/pmc_write_run.txt 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o k -- $CMD > $OUT/pmc_tcc_run.txt 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
