"""BASELINE.json configs[3] on one GPU: median ms per limo_ba_solve of the 10-keyframe / 8000-landmark window (LIMO_HIP_LIB selects the build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options(); c4 = synth.config_c4()
ts = []
for k in range(8):
    w = c4.copy(); t0 = time.perf_counter(); r = ctx.solve(w, o); ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
print("%-10s C4: median %.2f ms per solve, %d LM iterations, final cost %.6f" % (sys.argv[1] if len(sys.argv) > 1 else "", 1e3 * ts[len(ts) // 2], r["iterations_total"], r["final_cost"]))
