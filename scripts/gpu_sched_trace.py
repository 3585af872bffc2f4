"""Round structure of the streaming solve (KBA_SCHED_TRACE): slots in flight over the rounds, host time per round inside the enqueue
calls vs wall time per round.   usage: python scripts/gpu_sched_trace.py [batch ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
os.environ["KBA_SCHED_TRACE"] = "1"
from limo_amd import ba, default_options, synth

ctx = ba.Context(0)
o = default_options()
base = [synth.make_window(1000 + i, n_kf=5, n_lm=2000) for i in range(256)]
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    ws = [base[i % 256].copy() for i in range(n)]
    b = ba.Batch(ctx, ws)
    for rep in range(3):
        b.reset()
        t0 = time.perf_counter()
        b.solve(o)
        ctx.synchronize() if hasattr(ctx, "synchronize") else None
        print("batch %d: %.2f ms" % (n, 1e3 * (time.perf_counter() - t0)), flush=True)
    b.close()
