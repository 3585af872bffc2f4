"""Where limo_ba_batch_create spends its time (KBA_PACK_TRACE) for 1024 C2 host windows, at several KBA_PACK_THREADS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
from limo_amd import synth


def _make(seed):
    return synth.make_window(seed)


with mp.get_context("fork").Pool(32) as pool:
    ws = pool.map(_make, [7000 + i for i in range(1024)], chunksize=16)
from limo_amd import ba, default_options  # noqa: E402

ctx = ba.Context(0)
os.environ["KBA_PACK_TRACE"] = "1"
b = ba.Batch(ctx, [w.copy() for w in ws]); b.close()
for nt in sys.argv[1:] or ["16", "64"]:
    if nt == "default":
        os.environ.pop("KBA_PACK_THREADS", None)
    else:
        os.environ["KBA_PACK_THREADS"] = nt
    for rep in range(2):
        t0 = time.perf_counter()
        b = ba.Batch(ctx, [w for w in ws])
        t1 = time.perf_counter()
        b.close()
        sys.stderr.write("threads %s: create %.1f ms (python side)\n" % (nt, 1e3 * (t1 - t0)))
