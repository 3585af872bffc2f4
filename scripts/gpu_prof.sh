#!/bin/bash
# usage: gpu_prof.sh <tag> <batch>
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o $1 -- python scripts/gpu_sweep.py $2 > gpurun_out/prof_run.log 2>&1
tail -2 gpurun_out/prof_run.log | head -1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r01/$1_results.db')
for r in db.execute('select name,total_calls,total_duration,average,percentage from top_kernels'): print("%-70s %6d %12.1f %10.2f %6.2f"%(r[0][:70],r[1],r[2]/1e3,r[3]/1e3,r[4]))
PY
grep "B=" gpurun_out/prof_run.log
