"""Numbers quoted in BASELINE.md §4 that bench.py does not print: PCIe-inclusive BA rate, single-window latency,
C4-sized window, pose-only, depth estimator (C3) with its CPU oracle beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from limo_amd import ba, default_options, synth, synth_lidar
import pyoracle

ctx = ba.Context(0)
o = default_options()

# ---- PCIe-inclusive: pack + upload + solve + download of a batch handed over as host buffers
for B in (256, 1024):
    ws = [synth.make_window(3000 + i) for i in range(B)]
    b = ba.Batch(ctx, [w.copy() for w in ws]); b.solve(o); b.close()  # warm-up
    t0 = time.perf_counter()
    b = ba.Batch(ctx, ws)
    t1 = time.perf_counter()
    b.solve(o)
    t2 = time.perf_counter()
    b.download()
    t3 = time.perf_counter()
    b.close()
    print("PCIe-inclusive B=%d: create(pack+upload) %.1f ms, solve %.1f ms, download %.1f ms -> %.0f windows/s (resident: %.0f)"
          % (B, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), B / (t3 - t0), B / (t2 - t1)), flush=True)

# ---- one window at a time through limo_ba_solve (create + solve + download + destroy per call)
ws = [synth.make_window(3000 + i) for i in range(20)]
ctx.solve(ws[0].copy(), o)
t0 = time.perf_counter()
for w in ws:
    ctx.solve(w, o)
dt = (time.perf_counter() - t0) / len(ws)
print("limo_ba_solve (C2, one window per call, host buffers in/out): %.2f ms per window -> %.1f windows/s" % (dt * 1e3, 1 / dt), flush=True)

# ---- C4-sized single window (10 keyframes x 8000 landmarks), one GPU
c4 = synth.config_c4()
ctx.solve(c4.copy(), o)
t0 = time.perf_counter()
rep = ctx.solve(c4.copy(), o)
dt = time.perf_counter() - t0
print("C4 (10 KF x 8000 lm, %d obs) on ONE GPU: %.1f ms, %d LM iterations -> %.0f us per iteration" % (c4.n_obs, dt * 1e3, rep["iterations_total"], 1e6 * dt / max(1, rep["iterations_total"])), flush=True)
t0 = time.perf_counter()
wo = c4.copy()
pyoracle.load()
ro, _ = pyoracle.solve(wo, o, num_threads=3)
print("   oracle (3 threads): %.1f ms, %d iterations" % (1e3 * (time.perf_counter() - t0), ro["iterations_total"]), flush=True)

# ---- depth estimator, C3: 120k points, 1500 features
fr = synth_lidar.make_frame(1)
ba.depth_estimate(ctx, fr)
N = 50
t0 = time.perf_counter()
for _ in range(N):
    d = ba.depth_estimate(ctx, fr)
dt = (time.perf_counter() - t0) / N
pyoracle.load()
t0 = time.perf_counter()
for _ in range(3):
    d0 = pyoracle.depth_estimate(fr)
dtc = (time.perf_counter() - t0) / 3
print("depth C3 (%d points, %d features; host cloud in, host depths out): GPU %.2f ms/frame (%.0f frames/s), CPU oracle %.1f ms/frame; %d features with depth, max |diff| %.2e"
      % (fr["cloud"].shape[0], fr["uv"].shape[0], dt * 1e3, 1 / dt, dtc * 1e3, int((d > 0).sum()), float(np.abs(d - d0).max())), flush=True)
