"""Experiment: G independent sub-batches on G streams driven by G host threads vs one batch (same 1024 windows)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limo_amd import ba, default_options, synth

o = default_options()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = [synth.make_window(5000 + i) for i in range(N)]
for G in (1, 2, 4):
    ctxs = [ba.Context(0) for _ in range(G)]
    per = N // G
    bs = [ba.Batch(ctxs[g], [w.copy() for w in ws[g * per:(g + 1) * per]]) for g in range(G)]
    def run(b):
        b.reset(); b.solve(o)
    for b in bs: run(b)  # warm-up
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        th = [threading.Thread(target=run, args=(b,)) for b in bs]
        for t in th: t.start()
        for t in th: t.join()
    dt = (time.perf_counter() - t0) / reps
    st = [b.kernel_stats() for b in bs]
    print("G=%d: %.2f ms per %d windows -> %.0f windows/s; linearize ms (sum over groups) %.2f, schur ms %.2f" % (G, dt * 1e3, N, N / dt, sum(s["linearize_ms"] for s in st) / reps / 1, sum(s["schur_ms"] for s in st) / reps), flush=True)
    for b in bs: b.close()
