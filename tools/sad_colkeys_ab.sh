#!/bin/bash
# A/B builds of the SAD strip kernel's argmin: libofps_hip.so with -DOFPS_SAD_COLKEYS=0|1|2 under build/ab/colkeysN/
# (run HERE; hipcc cross-compiles), with the kernel's ISA beside each (-save-temps) for tools/sad_isa_stats.py.
# On the GPU box: tools/sad_colkeys_run.sh
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -fno-gpu-rdc"
python -m ofps_amd.build > /dev/null
for v in 0 1 2; do
    d=build/ab/colkeys$v
    rm -rf $d; mkdir -p $d
    $HIPCC $FLAGS -DOFPS_SAD_COLKEYS=$v -c ofps_amd/csrc/sad.hip -o $d/sad.o -save-temps=obj &
done
wait
for v in 0 1 2; do
    d=build/ab/colkeys$v
    objs=""
    for src in ofps_amd/csrc/*.hip; do
        stem=$(basename $src .hip)
        if [ $stem = sad ]; then objs="$objs $d/sad.o"; else objs="$objs ofps_amd/csrc/$stem.o"; fi
    done
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $d/libofps_hip.so $objs -Wl,-rpath,/opt/rocm/lib
    rm -f $d/*.bc $d/*.hipi $d/*.out $d/*.hipfb $d/*.resolution.txt $d/sad-host*
    for g in "8 32" "16 16"; do python tools/sad_isa_stats.py $d/sad-hip-amdgcn-amd-amdhsa-gfx950.s $g; done
done
