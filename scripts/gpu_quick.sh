#!/bin/bash
# quick check after a kernel change: GPU parity tests, one-iteration kernel averages (B=512), throughput sweep
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=512; ctx=ba.Context(0); o=default_options(max_num_iterations=1, num_trim_rounds=0)
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)])
for _ in range(6):
    b.reset(); b.solve(o)
PY
for st in ${STAGES:-0}; do
KBA_DEBUG_STAGE=$st rocprofv3 --kernel-trace --stats -d gpurun_out/prof_quick -o s$st -- python /tmp/one.py > gpurun_out/prof_run.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/prof_quick/s${st}_results.db')
for r in db.execute("select name,count(*),max(end-start)/1e3,avg(end-start)/1e3 from kernels where name like '%kba::%' group by name order by 3 desc"): print("stage $st %-40s %4d  max %9.1f us  avg %9.1f us"%(r[0][:40],r[1],r[2],r[3]))
PY
done
python scripts/gpu_sweep.py ${SIZES:-256 1024}
