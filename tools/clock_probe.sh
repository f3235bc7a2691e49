#!/bin/bash
# Shader clock while the cfg3 chain / the SAD bench step run: rocm-smi sampled every 50 ms for the whole life of the workload
# process; prints the histogram of the samples taken while the GPU was clocked up (> 1 GHz).  usage (GPU box): clock_probe.sh <out-file>
OUT=${1:-/dev/stdout}
cd $GRAFT_REPO_ROOT
probe() {   # $1 = label, rest = command
  local label=$1; shift
  "$@" > /dev/null 2>&1 &
  local P=$!
  local S=""
  while kill -0 $P 2>/dev/null; do S="$S $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | sed 's/.*(\([0-9]*\)Mhz).*/\1/')"; sleep 0.05; done
  echo "== $label"
  echo $S | tr ' ' '\n' | awk '$1 > 1000 {n++; s+=$1; if ($1 < mn || mn == 0) mn = $1; if ($1 > mx) mx = $1} END {if (n) printf "  %d samples above 1 GHz: min %d, mean %.0f, max %d MHz\n", n, mn, s / n, mx; else print "  no sample above 1 GHz"}'
}
{
probe "cfg3 chain (LK flow -> densify -> Almeida), 20000 iterations" python tools/prof_lk.py 20000
probe "SAD bench step, 1500 steps" python bench.py --no-cpu-baseline --no-end-to-end --no-legs --steps 1500
} > $OUT 2>&1
