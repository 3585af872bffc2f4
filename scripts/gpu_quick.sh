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
for r in db.execute("select name,total_calls,average from top_kernels where name like '%kba::%'"): print("stage $st %-40s %4d %9.1f us"%(r[0][:40],r[1],r[2]))
PY
done
python scripts/gpu_sweep.py ${SIZES:-256 1024}
