#!/bin/bash
# HBM traffic of k_schur / k_backsub / k_lm_accum (full-batch launches, B=512 one-iteration profile), separate --pmc passes
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=512; ctx=ba.Context(0); o=default_options(max_num_iterations=1, num_trim_rounds=0)
ws=[synth.make_window(5000+i) for i in range(B)]
print("obs", sum(w.n_obs for w in ws), "landmarks", sum(w.n_lm for w in ws))
b=ba.Batch(ctx,ws)
for _ in range(3):
    b.reset(); b.solve(o)
PY
for ctr in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc_s_$ctr -o s -- python /tmp/one.py > gpurun_out/pmc_s_$ctr.log 2>&1
grep "^obs" gpurun_out/pmc_s_$ctr.log
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/pmc_s_$ctr/s_results.db')
for r in db.execute("select name, count(*), max(counter_value) from pmc_events where name like '%k_schur%' or name like '%k_backsub%' or name like '%k_lm_accum%' or name like '%k_cost%' or name like '%k_cam_solve%' group by name"): print("$ctr %-34s %3d max %.0f KiB"%(r[0].split('(')[0][-34:], r[1], r[2]))
PY
done
