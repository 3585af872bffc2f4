#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ba.py -m gpu -x -q 2>&1 | tail -5
python scripts/gpu_sweep.py 1 8 64 256 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o sweep64e -- python scripts/gpu_sweep.py 64 > gpurun_out/prof_run.log 2>&1
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r01/sweep64e_results.db')
for r in db.execute('select name,total_calls,total_duration,average,percentage from top_kernels'): print("%-70s %6d %12.1f %10.2f %6.2f"%(r[0][:70],r[1],r[2]/1e3,r[3]/1e3,r[4]))
PY
