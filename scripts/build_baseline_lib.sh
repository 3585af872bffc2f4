#!/bin/bash
# Builds liblimo_hip.so as it was at git revision REV into limo_amd/lib/variants/liblimo_hip_<TAG>.so (default TAG = base): the
# reference build of the per-kernel regression gate (scripts/kernel_gate.py, scripts/gpu_round_r06.sh) - the previous round's kernels
# timed on the SAME box as this round's.  The sources are checked out into a scratch directory, nothing in the work tree changes.
#   usage: scripts/build_baseline_lib.sh REV [TAG]
set -e
cd "$(dirname "$0")/.."
rev=$1; tag=${2:-base}
tmp=$(mktemp -d /tmp/limo_base.XXXXXX)
git archive "$rev" limo_amd/csrc include | tar -x -C "$tmp"
mkdir -p limo_amd/lib/variants "$tmp/obj"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w"
pids=()
for src in limo_hip.hip kba_pack.cpp host_misc.cpp depth.hip landmark_init.hip; do
  extra=""; case $src in depth.hip|landmark_init.hip) extra="-ffp-contract=off";; esac
  ( cd "$tmp/limo_amd/csrc" && hipcc $flags $extra -c -o "$tmp/obj/$src.o" $src ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -fPIC -shared -o limo_amd/lib/variants/liblimo_hip_$tag.so "$tmp"/obj/*.o -pthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$tmp"
echo "$rev" > limo_amd/lib/variants/liblimo_hip_$tag.rev
ls -la limo_amd/lib/variants/liblimo_hip_$tag.so
