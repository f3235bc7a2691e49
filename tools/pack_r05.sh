#!/bin/bash
# gpurun_out/r05p (written by tools/collect_r05.sh on the GPU box) -> the committed form under profiles/r05/
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r05p; D=profiles/r05
mkdir -p $D
for p in sad_strip sad_strip_cfg4 cfg3_chain cfg3_chain_pm16; do
  python tools/pack_profile.py $S/$p $D/$p
  for f in summary.json stage_times.json; do [ -f $S/$p/$f ] && cp $S/$p/$f $D/$p/; done
done
mkdir -p $D/cfg5_stream && cp $S/cfg5_stream/*.json $D/cfg5_stream/
cp $S/*.json $S/*.txt $D/
mv $D/accuracy_table.txt $D/accuracy.txt          # the name VERDICT r4 item 3 gave it
rm -f $D/hbm_traffic.json
python tools/make_hbm_traffic.py
cmp -s profiles/hbm_traffic.json $S/hbm_traffic.json || echo "note: hbm_traffic.json differs from the one the bench line of this collection read"
{ for f in lk farneback almeida sad densify detect mask pipeline multi; do python tools/kernel_resources.py ofps_amd/csrc/$f.hip 2>/dev/null | sed "s/^/$f.hip  /"; done; } > $D/kernel_resources.txt
# (round 6: the gate reads profiles/r05/bench_n1.json itself; no refreshed copy)
