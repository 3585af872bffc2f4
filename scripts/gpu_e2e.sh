#!/bin/bash
# end_to_end of the bench (pack + upload + solve + download of 1024 host windows) a few times, with the pack / create trace.
KBA_PACK_TRACE=1 python - <<'PY' 2>&1 | grep -E "\[kba\]|e2e" | tail -30
import os, sys, time
sys.path.insert(0, os.getcwd())
import multiprocessing as mp
from limo_amd import synth
def _make(seed): return synth.make_window(seed)
with mp.get_context("fork").Pool(32) as pool:
    ws = pool.map(_make, [7000 + i for i in range(1024)], chunksize=16)
from limo_amd import ba, default_options
from limo_amd.window import struct_array
ctx = ba.Context(0); o = default_options()
b = ba.Batch(ctx, [w.copy() for w in ws]); b.solve(o); b.download(); b.close()
for rep in range(4):
    cur = [w.copy() for w in ws]
    arr = struct_array(cur)
    t0 = time.perf_counter(); b = ba.Batch(ctx, cur, arr); t1 = time.perf_counter(); b.solve(o); t2 = time.perf_counter(); b.download(); t3 = time.perf_counter(); b.close()
    sys.stderr.write("e2e: create %.1f solve %.1f download %.1f ms -> %.0f windows/s\n" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1024/(t3-t0)))
PY
