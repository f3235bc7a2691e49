#!/bin/bash
# kernel + copy trace of one steady-state frame of the hip_flow read-ahead decoder, this tree and build/r05tree (tools/ab_r05_r06.sh says how to make it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4; mkdir -p $O
for tree in r06 r05; do
  if [ $tree = r05 ]; then D=$R/build/r05tree; else D=$R; fi
  [ -d $D ] || continue
  (cd $D && for i in 1 2 3; do python $R/tools/fb_decoder_trace.py 200; done) > $O/fb_dec_time_$tree.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fbd_$tree && cd $D && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fbd_$tree -o k -- python $R/tools/fb_decoder_trace.py 40 > /dev/null 2>&1)
  python - $tree <<'PY' > $O/fb_dec_trace_$tree.txt
import csv, glob, sys
tree = sys.argv[1]
ev = []
for f in glob.glob(f"/tmp/fbd_{tree}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
for f in glob.glob(f"/tmp/fbd_{tree}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
# one steady-state frame: from the 30th H2D copy to the 31st
copies = [i for i, e in enumerate(ev) if e[2].startswith("COPY MEMORY_COPY_HOST_TO_DEVICE")]
a, b = copies[-6], copies[-5]
t0 = ev[a][0]
busy = 0
for s, e, n, q in ev[a:b]:
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f} us  q{q:>3s}  {n}")
print(f"frame period {(ev[b][0]-t0)/1e3:.1f} us, {b-a} events, sum of durations {sum(e-s for s,e,_,_ in ev[a:b])/1e3:.1f} us")
PY
done
cat $O/fb_dec_time_r06.txt $O/fb_dec_time_r05.txt
