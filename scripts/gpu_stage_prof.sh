#!/bin/bash
# k_cam_solve / k_cam_assemble stage timings via the KBA_DEBUG_STAGE early exits, one LM iteration, B in $SIZES
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=int(sys.argv[1]); ctx=ba.Context(0); o=default_options(max_num_iterations=1, num_trim_rounds=0)
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)])
for _ in range(6):
    b.reset(); b.solve(o)
PY
for B in ${SIZES:-1 512}; do
for st in ${STAGES:-0}; do
KBA_DEBUG_STAGE=$st rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stage -o s$st -- python /tmp/one.py $B > gpurun_out/prof_run.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/prof_stage/s${st}_results.db')
for r in db.execute("select name,count(*),max(end-start)/1e3,avg(end-start)/1e3 from kernels where name like '%${KERNEL:-k_cam}%' group by name order by 3 desc"): print("B=$B stage $st %-40s %4d  max %9.1f us  avg %9.1f us"%(r[0][:40],r[1],r[2],r[3]))
PY
done
done
