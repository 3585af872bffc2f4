#!/bin/bash
# Builds liblimo_hip variants with -DKBA_ABLATE=<n> (pieces of k_lin_lm left out; results wrong by design) into
# limo_amd/lib/ablate/liblimo_hip_<n>.so - for scripts/gpu_lin_ablate.sh.   usage: scripts/build_ablations.sh 61 62 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p limo_amd/lib/ablate
build_one() {
  n=$1; d=limo_amd/lib/ablate/obj_$n; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DKBA_ABLATE=$n -c -o $d/limo_hip.o limo_amd/csrc/limo_hip.hip
  hipcc --offload-arch=gfx950 -fPIC -shared -o limo_amd/lib/ablate/liblimo_hip_$n.so $d/limo_hip.o limo_amd/lib/obj/kba_pack.cpp.o limo_amd/lib/obj/host_misc.cpp.o limo_amd/lib/obj/depth.hip.o limo_amd/lib/obj/landmark_init.hip.o -pthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  rm -rf $d
}
for n in "$@"; do build_one $n & done
wait
ls -la limo_amd/lib/ablate
