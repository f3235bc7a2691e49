cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ps -o k -- python $GRAFT_REPO_ROOT/tools/prof_stream.py 60 > /dev/null 2>&1; python - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/ps/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-42:]))
for f in glob.glob("/tmp/ps/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
copies = [i for i, e in enumerate(ev) if e[2].startswith("COPY")]
start = copies[-4]
t0 = ev[start][0]
for s, e, n in ev[start:start+30]:
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f} us  {n}")
PY
