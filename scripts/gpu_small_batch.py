"""Small HBM-resident batches (B in {1, 64, 1024}): reset + solve per batch size on the one-launch path
(k_solve_coop with 256 / B workgroups per window) and on the streaming solve (KBA_STREAM_MIN=1)."""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth

ctx = ba.Context(0)
o = default_options()
base = [synth.make_window(3000 + i) for i in range(512)]
for B in (1, 4, 16, 32, 64, 128, 256, 512):
    row = []
    for mode in ("one launch", "streaming"):
        os.environ.pop("KBA_STREAM_MIN", None)
        if mode == "streaming":
            os.environ["KBA_STREAM_MIN"] = "1"
        b = ba.Batch(ctx, [w.copy() for w in base[:B]])
        b.reset(); b.solve(o)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); b.reset(); b.solve(o); ts.append(time.perf_counter() - t0)
        b.close()
        row.append(statistics.median(ts))
    os.environ.pop("KBA_STREAM_MIN", None)
    print("B = %4d: default path %8.2f ms (%7.0f windows/s) | streaming solve %8.2f ms (%7.0f windows/s)" % (B, row[0] * 1e3, B / row[0], row[1] * 1e3, B / row[1]))
