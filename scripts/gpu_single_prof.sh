#!/bin/bash
# One C2 window per call through limo_ba_solve: wall time per call, and from the kernel trace the device-busy time
# and the idle gaps between kernels of one call (launch-latency bound or kernel bound?).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/single.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options()
ws = [synth.make_window(3000 + i) for i in range(12)]
ctx.solve(ws[0].copy(), o); ctx.solve(ws[1].copy(), o)
t0 = time.perf_counter(); its = 0
for w in ws[2:]:
    its += ctx.solve(w, o)["iterations_total"]
dt = (time.perf_counter() - t0) / 10
print("single: %.2f ms per window, %.1f LM iterations per window -> %.0f us per iteration" % (dt * 1e3, its / 10, 1e6 * dt * 10 / its))
PY
python /tmp/single.py
KBA_NO_COOP_SOLVE=1 python /tmp/single.py | sed 's/^single:/single [lock-step launches, KBA_NO_COOP_SOLVE=1]:/'
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_single -o s -- python /tmp/single.py > gpurun_out/prof_single.log 2>&1
grep "^single" gpurun_out/prof_single.log
python scripts/prof_summary.py gpurun_out/prof_single/s_results.db | head -24
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect("gpurun_out/prof_single/s_results.db")
rows = db.execute("select start,end from kernels order by start").fetchall()
busy = sum(e - s for s, e in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g < 200000]
print("kernels %d, busy %.1f ms, gaps<200us: %d, sum %.1f ms, median %.2f us, mean %.2f us" % (len(rows), busy / 1e6, len(small), sum(small) / 1e6, sorted(small)[len(small) // 2] / 1e3, sum(small) / len(small) / 1e3))
PY
