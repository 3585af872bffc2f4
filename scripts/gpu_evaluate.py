"""k_evaluate alone: parity against the oracle (C2, with and without the loss) and its device time over 1024 C2 windows
(the `roofline_evaluate` entry of bench.py).   usage: python scripts/gpu_evaluate.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import pyoracle
from limo_amd import ba, default_options, synth

ctx = ba.Context(0)
o = default_options()
w = synth.config_c2()
for apply_loss in (True, False):
    c0, r0, jp0, jl0, v0 = pyoracle.evaluate(w, o, apply_loss)
    c1, r1, jp1, jl1, v1 = ctx.evaluate(w, o, apply_loss)
    print("apply_loss=%d: valid equal %s, rel cost %.2e, r %.2e, Jp %.2e, Jl %.2e" % (
        apply_loss, np.array_equal(v0, v1), abs(c0 - c1) / abs(c0), np.abs(r0 - r1).max() / max(1.0, np.abs(r0).max()),
        np.abs(jp0 - jp1).max() / np.abs(jp0).max(), np.abs(jl0 - jl1).max() / np.abs(jl0).max()))
ws = [synth.make_window(1000 + i, n_kf=5, n_lm=2000) for i in range(256)]
ws = (ws * 4)[:1024]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    ms, n_obs, n_dep = ba.evaluate_batch_time(ctx, ws, o, reps=reps)
    alg = 212 * n_obs + 84 * n_dep
    print("k_evaluate: %.3f ms for %d observations (%d with depth): %.0f GB/s on the algorithmic bytes = %.3f of 8 TB/s" % (
        ms, n_obs, n_dep, alg / (ms * 1e-3) / 1e9, alg / (ms * 1e-3) / 8e12))
