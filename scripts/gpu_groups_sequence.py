"""Do batches of different sizes solved one after the other on ONE context disturb each other (slot groups, pooled blocks)?
   python scripts/gpu_groups_sequence.py B1 B2 ...   (KBA_GROUPS from the environment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from limo_amd import synth
def _make(seed): return synth.make_window(seed)
with mp.get_context("fork").Pool(32) as pool:
    ws = pool.map(_make, [7000 + i for i in range(1024)], chunksize=16)
from limo_amd import ba, default_options
ctx = ba.Context(0); o = default_options()
for B in [int(a) for a in sys.argv[1:]]:
    cur = [ws[i % 1024].copy() for i in range(B)]
    b = ba.Batch(ctx, cur)
    ts = []
    for i in range(3):
        if i: b.reset()
        t0 = time.perf_counter(); b.solve(o); ts.append(1e3 * (time.perf_counter() - t0))
    b.close()
    print("B=%5d  solves: %s ms" % (B, " ".join("%.1f" % t for t in ts)))
