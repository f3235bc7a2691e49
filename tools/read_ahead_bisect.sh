#!/bin/bash
# Builds A/B variants of libofps_hip.so for the read-ahead regression hunt under build/ab/<name>/
# (git-ignored, travels to the GPU box) -- run HERE (hipcc cross-compiles), then on the GPU box:
#   python tools/read_ahead_probe.py --all > gpurun_out/read_ahead_bisect.txt
# Variants:  product   = the tree as it is (both kinds of page-locked block hipHostMallocCoherent)
#            default   = hipHostMallocDefault for both (what round 3 shipped)
#            noncoh    = hipHostMallocNonCoherent for both (coarse-grained: kernel stores visible at kernel end)
#            spin5ms   = product flags, frame_wait polls for 5 ms before it blocks
#            nospin    = product flags, frame_wait blocks at once (hipEventSynchronize)
#            r03       = the round-3 tree (git archive cae37ba), built with its own flags
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -fno-gpu-rdc"
variant() {   # name, extra -D flags: only the translation units that look at the constants are rebuilt
    local name=$1; shift
    local d=build/ab/$name
    mkdir -p $d
    local objs=""
    for src in ofps_amd/csrc/*.hip; do
        stem=$(basename $src .hip)
        case $stem in
            ctx|pipeline|lk) $HIPCC $FLAGS "$@" -c $src -o $d/$stem.o & objs="$objs $d/$stem.o" ;;
            *) objs="$objs ofps_amd/csrc/$stem.o" ;;
        esac
    done
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $d/libofps_hip.so $objs -Wl,-rpath,/opt/rocm/lib
    echo "built $d/libofps_hip.so"
}
python -m ofps_amd.build >/dev/null
variant default -DOFPS_HIP_HOST_BLOCK_FLAGS=hipHostMallocDefault -DOFPS_HIP_HOST_USER_FLAGS=hipHostMallocDefault
variant noncoh  -DOFPS_HIP_HOST_BLOCK_FLAGS=hipHostMallocNonCoherent -DOFPS_HIP_HOST_USER_FLAGS=hipHostMallocNonCoherent
variant spin5ms -DOFPS_HIP_FRAME_WAIT_SPIN_US=5000
variant nospin  -DOFPS_HIP_FRAME_WAIT_SPIN_US=0
if [ "$1" = "--r03" ]; then
    d=build/ab/r03
    rm -rf $d; mkdir -p $d/src
    git archive cae37ba ofps_amd/csrc include | tar -x -C $d/src
    objs=""
    for src in $d/src/ofps_amd/csrc/*.hip; do
        stem=$(basename $src .hip)
        $HIPCC $FLAGS -c $src -o $d/$stem.o & objs="$objs $d/$stem.o"
    done
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $d/libofps_hip.so $objs -Wl,-rpath,/opt/rocm/lib
    rm -rf $d/src
    echo "built $d/libofps_hip.so"
fi
