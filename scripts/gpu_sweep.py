"""Timing sweep over batch sizes (no torch): per-step wall time, device time, linearize kernel share."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limo_amd import ba, default_options, synth

ctx = ba.Context(0)
opts = default_options()
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256]
pool = [synth.make_window(5000 + i) for i in range(max(sizes))]
for B in sizes:
    b = ba.Batch(ctx, [w.copy() for w in pool[:B]])
    b.solve(opts)  # warm-up
    b.kernel_stats(reset=True)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        b.reset()
        b.solve(opts)
    dt = (time.perf_counter() - t0) / n
    st = b.kernel_stats()
    reps = b.download()
    its = [r["iterations_total"] for r in reps]
    print("B=%4d  step %.2f ms  -> %.1f windows/s | device %.2f ms/step, linearize %.2f ms/step over %d launches | iters mean %.1f max %d, converged %d/%d"
          % (B, dt * 1e3, B / dt, st["total_ms"] / n, st["linearize_ms"] / n, st["linearize_launches"] / n, np.mean(its), max(its), sum(r["termination"] == 0 for r in reps), B), flush=True)
    b.close()
