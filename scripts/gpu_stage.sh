#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for st in 1 2 3 0; do
KBA_DEBUG_STAGE=$st rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stage -o st$st -- python scripts/gpu_sweep.py 64 > gpurun_out/prof_run.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/prof_stage/st${st}_results.db')
for r in db.execute("select name,total_calls,total_duration,average from top_kernels where name like '%cam_solve%' or name like '%cam_assemble%'"): print("stage $st %-50s %6d %10.1f %8.2f"%(r[0][:50],r[1],r[2]/1e3,r[3]/1e3))
PY
done
