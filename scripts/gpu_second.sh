#!/bin/bash
set -x
mkdir -p gpurun_out
python scripts/gpu_sweep.py 1 8 64 256 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o sweep64 -- python scripts/gpu_sweep.py 64 > gpurun_out/prof_run.log 2>&1
tail -3 gpurun_out/prof_run.log
find gpurun_out/prof_r01 -name "*stats*" | head
for f in $(find gpurun_out/prof_r01 -name "*kernel_stats*.csv"); do head -30 $f; done
