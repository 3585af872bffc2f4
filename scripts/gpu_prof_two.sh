#!/bin/bash
# kernel table (one slot group) of two library builds on one box: scripts/gpu_prof_two.sh TAG1:LIB1 TAG2:LIB2  (LIB = "" for the product build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof2
for cfg in "$@" "$1"; do
  tag=${cfg%%:*}; lib=${cfg#*:}
  rm -rf gpurun_out/prof2/$tag
  env KBA_GROUPS=1 ${lib:+LIMO_HIP_LIB=$lib} timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof2/$tag -o bench -- python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/prof2/$tag.log 2>&1
  echo "== $tag"; python scripts/prof_summary.py gpurun_out/prof2/$tag/bench_results.db | head -8
  rm -rf gpurun_out/prof2/$tag
done
