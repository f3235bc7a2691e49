#!/bin/bash
# Shader clock while a kernel mix runs: samples rocm-smi every 50 ms beside `python tools/prof_lk.py N` (cfg3 chain) and beside
# the SAD bench step.  usage (GPU box): clock_probe.sh <out-file>
OUT=${1:-/dev/stdout}
cd $GRAFT_REPO_ROOT
sample() { for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.05; done; }
{
echo "== idle"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3
echo "== during the cfg3 chain (LK flow -> densify -> Almeida), 4000 iterations"
python tools/prof_lk.py 4000 > /dev/null 2>&1 &
P=$!; sleep 1.0; sample | sort | uniq -c; wait $P
echo "== during the SAD bench step"
python bench.py --no-cpu-baseline --no-end-to-end --no-legs --steps 400 > /dev/null 2>&1 &
P=$!; sleep 6.0; sample | sort | uniq -c; wait $P
} > $OUT 2>&1
