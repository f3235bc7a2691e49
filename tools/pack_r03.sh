#!/bin/bash
# gpurun_out/r03p (written by tools/collect_r03.sh on the GPU box) -> the committed form under profiles/r03/
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r03p; D=profiles/r03
for p in sad_strip sad_strip_cfg4 cfg3_chain; do
  python tools/pack_profile.py $S/$p $D/$p
  for f in summary.json stage_times.json; do [ -f $S/$p/$f ] && cp $S/$p/$f $D/$p/; done
done
mkdir -p $D/cfg5_stream && cp $S/cfg5_stream/*.json $D/cfg5_stream/
cp $S/*.json $S/*.txt $D/
cp $S/ubench_lds.txt profiles/ubench_lds_r03.txt; cp $S/ubench_valu.txt profiles/ubench_valu_r03.txt
cp $S/ubench_allgather.txt profiles/ubench_allgather_r03.txt
rm -f $D/ubench_lds.txt $D/ubench_valu.txt $D/ubench_allgather.txt $D/hbm_traffic.json
python tools/make_hbm_traffic.py
cmp -s profiles/hbm_traffic.json $S/hbm_traffic.json || echo "note: hbm_traffic.json differs from the one the bench line of this collection read"
