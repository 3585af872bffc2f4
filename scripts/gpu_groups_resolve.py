import os, sys, time
sys.path.insert(0, os.getcwd())
import multiprocessing as mp
from limo_amd import synth
def _make(seed): return synth.make_window(seed)
with mp.get_context("fork").Pool(32) as pool:
    ws = pool.map(_make, [7000 + i for i in range(1024)], chunksize=16)
from limo_amd import ba, default_options
ctx = ba.Context(0); o = default_options()
B = 1024
b = ba.Batch(ctx, [w.copy() for w in ws[:B]])
for i in range(4):
    if i: b.reset()
    t0 = time.perf_counter(); b.solve(o); t = time.perf_counter() - t0
    sys.stderr.write("solve %d: %.2f ms\n" % (i, 1e3 * t))
b.close()
b = ba.Batch(ctx, [w.copy() for w in ws[:B]])
t0 = time.perf_counter(); b.solve(o); t = time.perf_counter() - t0
sys.stderr.write("fresh batch, same context: %.2f ms\n" % (1e3 * t))
