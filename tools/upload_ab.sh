#!/bin/bash
# A/B builds of the per-frame / batched upload (profiles/r05/batched_bimodal.txt): build/ab/up_<name>/{libofps_hip.so,ofps_hip_tool}
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -fno-gpu-rdc"
python -m ofps_amd.build > /dev/null
variant() {
    local name=$1; shift
    local d=build/ab/up_$name
    mkdir -p $d
    $HIPCC $FLAGS "$@" -c ofps_amd/csrc/pipeline.hip -o $d/pipeline.o
    objs=""
    for src in ofps_amd/csrc/*.hip; do
        stem=$(basename $src .hip)
        if [ $stem = pipeline ]; then objs="$objs $d/pipeline.o"; else objs="$objs ofps_amd/csrc/$stem.o"; fi
    done
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $d/libofps_hip.so $objs -Wl,-rpath,/opt/rocm/lib
    cp ofps_amd/host/ofps_hip_tool $d/
    echo built $d
}
# dma: every upload through hipMemcpyAsync (rounds 1-4); kN: batched AND single-frame uploads through an N-workgroup copy kernel
# (the product: batched through 16 workgroups, single-frame through the DMA engine)
variant dma -DOFPS_HIP_UPLOAD_WGS=0
for n in 16 32 64 128; do variant k$n -DOFPS_HIP_UPLOAD_WGS=$n -DOFPS_HIP_UPLOAD_KERNEL_SINGLE=1; done
variant product
