"""Micro-benchmark of the Jacobian-evaluation kernel on a full batch: every window linearises exactly once per solve
(max_num_iterations = 0, no trimming), so HIP-event time / launches is the full-batch k_linearize time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limo_amd import ba, default_options, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = ba.Context(0)
opts = default_options(max_num_iterations=0, num_trim_rounds=0)
ws = [synth.make_window(7000 + i) for i in range(B)]
b = ba.Batch(ctx, ws)
n_obs = sum(w.n_obs for w in ws)
n_dep = int(sum((w.obs_d > 0).sum() for w in ws))
alg = 212 * n_obs + 84 * n_dep  # SURVEY 8d work unit (materialised Jacobian); stored: 52 B read + 56 B written
stored = 108 * n_obs
for _ in range(3):
    b.reset(); b.solve(opts)
b.kernel_stats(reset=True)
N = 20
for _ in range(N):
    b.reset(); b.solve(opts)
st = b.kernel_stats()
# every solve launches k_linearize twice here (the second launch finds nothing to linearise and exits at once):
# time per solve = the full-batch launch (+ ~5 us of the empty one)
ms = st["linearize_ms"] / N
print("B=%d obs=%d  k_linearize %.1f us/launch  algorithmic (SURVEY 8d) %.1f MB -> %.0f GB/s (%.2f of 8 TB/s); stored bytes %.1f MB -> %.0f GB/s  [variant %s]" % (B, n_obs, ms * 1e3, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000, stored / 1e6, stored / ms / 1e6, os.environ.get("KBA_DEBUG_STAGE", "0")))
