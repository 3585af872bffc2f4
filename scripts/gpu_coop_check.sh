#!/bin/bash
# One-launch solve of a single window (k_solve_coop): bit-equality with the lock-step launches, latency per limo_ba_solve on
# both paths (no profiler attached), then the kernel table.  Every step under its own timeout (a barrier that is not met
# aborts the launch after 0.5 s by itself).
OUT=gpurun_out/coop
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/single.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options()
ws = [synth.make_window(3000 + i) for i in range(12)]
ctx.solve(ws[0].copy(), o); ctx.solve(ws[1].copy(), o)
ts = []; its = 0
for w in ws[2:]:
    t0 = time.perf_counter(); its += ctx.solve(w, o)["iterations_total"]; ts.append(time.perf_counter() - t0)
dt = sum(ts) / len(ts); ts.sort()
print("single [%s]: %.2f ms per window (median %.2f), %.1f LM iterations per window -> %.0f us per iteration" % (
    "lock-step launches" if os.environ.get("KBA_NO_COOP_SOLVE") else "one launch, G=%s" % os.environ.get("KBA_COOP_G", "auto"),
    dt * 1e3, ts[len(ts) // 2] * 1e3, its / 10, 1e6 * dt * 10 / its))
PY
timeout 120 python /tmp/single.py || echo "single.py failed / timed out"
KBA_NO_COOP_SOLVE=1 timeout 120 python /tmp/single.py
timeout 900 python -m pytest tests/test_gpu_ba.py -x -q -m gpu -k "one_launch" 2>&1 | tail -8
for g in 4 6 12 16; do KBA_COOP_G=$g timeout 120 python /tmp/single.py; done
KBA_COOP_PLAIN_LAUNCH=1 timeout 120 python /tmp/single.py
KBA_HOST_TRACE=1 timeout 120 python /tmp/single.py 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python /tmp/single.py > $OUT/prof.log 2>&1
python scripts/prof_summary.py $OUT/prof/s_results.db | head -8
rm -rf $OUT/prof
