#!/bin/bash
# gpurun_out/r06p (written by tools/collect_r06.sh on the GPU box) -> the committed form under profiles/r06/
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r06p; D=profiles/r06
mkdir -p $D
for p in sad_strip sad_strip_cfg4 cfg3_chain cfg3_chain_pm16; do
  python tools/pack_profile.py $S/$p $D/$p
  for f in summary.json stage_times.json; do [ -f $S/$p/$f ] && cp $S/$p/$f $D/$p/; done
done
cp $S/*.json $S/*.txt $D/
mv $D/accuracy_table.txt $D/accuracy.txt
rm -f $D/hbm_traffic.json
python tools/make_hbm_traffic.py
cmp -s profiles/hbm_traffic.json $S/hbm_traffic.json || echo "note: hbm_traffic.json differs from the one the bench line of this collection read"
{ for f in lk farneback almeida sad densify detect mask frontend pipeline multi; do python tools/kernel_resources.py ofps_amd/csrc/$f.hip 2>/dev/null | sed "s/^/$f.hip  /"; done; } > $D/kernel_resources.txt
python tools/almeida_spill_static.py > $D/almeida_spill_static.txt
