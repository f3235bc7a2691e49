#!/bin/bash
# Runs the default bench line N times (fresh process each) and prints the spread: evidence that the headline is stable.
N=${1:-20}
for i in $(seq 1 $N); do python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline 2>/dev/null | tail -1; done | python3 -c "
import sys, json
v=[json.loads(l) for l in sys.stdin if l.strip().startswith('{')]
vals=[d['value'] for d in v]; lm=[d['roofline']['launch_ms'] for d in v]
print(json.dumps({'runs': len(vals), 'Mvectors_per_s': {'min': min(vals), 'max': max(vals), 'mean': round(sum(vals)/len(vals),2)},
                  'launch_ms': {'min': min(lm), 'max': max(lm)}, 'parity_ok': all(d['parity_check']['ok'] for d in v), 'values': vals}))"
