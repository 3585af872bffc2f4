#!/bin/bash
# SQ counters of the one-iteration profile (B=512): instruction mix / stall picture per kernel
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=512; ctx=ba.Context(0); o=default_options(max_num_iterations=1, num_trim_rounds=0)
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)])
for _ in range(3):
    b.reset(); b.solve(o)
PY
if [ "$1" == "list" ]; then rocprofv3 --list-avail 2>/dev/null | grep -i "^\s*Name\|SQ_\|GRBM" | head -150; exit 0; fi
i=0
for grp in "$@"; do
i=$((i+1))
rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_sq -o g$i -- python /tmp/one.py > gpurun_out/pmc_sq_$i.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/pmc_sq/g${i}_results.db')
rows=db.execute("select name, counter_name, max(counter_value), max(duration) from pmc_events where name like '%k_schur%' or name like '%k_linearize%' or name like '%k_backsub%' group by name, counter_name").fetchall()
for r in rows: print("%-28s %-28s %14.0f  (max dur %.0f us)"%(r[0].split('(')[0][-28:], r[1], r[2], r[3]/1e3))
PY
done
