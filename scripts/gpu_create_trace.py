"""Where the host time of limo_ba_batch_create goes for 1024 C2 windows (KBA_PACK_TRACE): pack (fill / merge) vs upload, five calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
os.environ["KBA_PACK_TRACE"] = "1"
from limo_amd import ba, default_options, synth
from limo_amd.window import struct_array

ctx = ba.Context(0)
base = [synth.make_window(1000 + i, n_kf=5, n_lm=2000) for i in range(256)]
ws = [base[i % 256].copy() for i in range(1024)]
arr = struct_array(ws)
for rep in range(5):
    t0 = time.perf_counter()
    b = ba.Batch(ctx, ws, arr)
    t1 = time.perf_counter()
    print("create %.1f ms" % (1e3 * (t1 - t0)), flush=True)
    b.close()
