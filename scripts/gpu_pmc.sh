#!/bin/bash
# HBM traffic counters of k_linearize (separate --pmc passes, kernel-trace only), full-batch micro-benchmark
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc_$ctr -o lin -- python scripts/gpu_lin_bench.py 256 > gpurun_out/pmc_$ctr.log 2>&1
tail -1 gpurun_out/pmc_$ctr.log
python - <<PY
import sqlite3, glob
f=glob.glob('gpurun_out/pmc_$ctr/*results.db')[0]
db=sqlite3.connect(f)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand=[t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()]
print(cand[:12])
for t in cand:
    try:
        cols=[r[1] for r in db.execute(f"pragma table_info({t})")]
        n=db.execute(f"select count(*) from {t}").fetchone()[0]
        print(t, n, cols[:14])
    except Exception as e: print(t, e)
PY
done
