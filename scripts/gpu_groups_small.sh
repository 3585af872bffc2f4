#!/bin/bash
# slot groups (KBA_GROUPS) for small and medium resident batches: reset + solve of B C2 windows, ms per solve
cat > /tmp/gs.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import multiprocessing as mp
from limo_amd import synth
def _make(seed): return synth.make_window(seed)
with mp.get_context("fork").Pool(32) as pool:
    ws = pool.map(_make, [7000 + i for i in range(1024)], chunksize=16)
from limo_amd import ba, default_options
ctx = ba.Context(0); o = default_options()
for B in (32, 64, 128, 256, 512, 1024):
    b = ba.Batch(ctx, [w.copy() for w in ws[:B]])
    b.solve(o)
    ts = []
    for _ in range(3):
        b.reset(); t0 = time.perf_counter(); b.solve(o); ts.append(time.perf_counter() - t0)
    b.close()
    print("B=%4d  %.2f ms per solve  %.0f windows/s" % (B, 1e3 * min(ts), B / min(ts)))
PY
for g in 1 2; do echo "== KBA_GROUPS=$g"; KBA_GROUPS=$g python /tmp/gs.py 2>&1 | grep "B="; done
