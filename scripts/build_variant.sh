#!/bin/bash
# Builds a variant of liblimo_hip.so with extra compiler flags into limo_amd/lib/variants/liblimo_hip_<TAG>.so (selected at run time
# with LIMO_HIP_LIB=...; travels to the GPU box with the snapshot).   usage: scripts/build_variant.sh TAG "-DKBA_CAM_SOLVE_WAVES=4 ..."
set -e
cd "$(dirname "$0")/.."
tag=$1; flags=$2
mkdir -p limo_amd/lib/variants
d=limo_amd/lib/variants/obj_$tag; mkdir -p $d
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c -o $d/limo_hip.o limo_amd/csrc/limo_hip.hip 2>&1 | grep -i "error" || true
hipcc --offload-arch=gfx950 -fPIC -shared -o limo_amd/lib/variants/liblimo_hip_$tag.so $d/limo_hip.o limo_amd/lib/obj/kba_pack.cpp.o limo_amd/lib/obj/host_misc.cpp.o limo_amd/lib/obj/depth.hip.o limo_amd/lib/obj/landmark_init.hip.o -pthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf $d
ls -la limo_amd/lib/variants/liblimo_hip_$tag.so
