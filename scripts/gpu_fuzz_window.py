"""One window of the fuzz sweep (tests/fuzz_common.py:random_windows) solved on the GPU, its report next to the oracle's:
python scripts/gpu_fuzz_window.py SEED N INDEX [TAG]   (LIMO_HIP_LIB selects the library build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import fuzz_common as fc
from limo_amd import ba, default_options
seed, n, idx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "gpu"
kw, w = fc.random_windows(n, seed)[idx]
o = default_options()
ctx = ba.Context(0)
wg = w.copy()
r = ctx.solve(wg, o)
keys = ("termination", "iterations_total", "successful_steps", "num_linearizations", "final_cost", "n_trimmed_landmarks")
print("%-10s" % tag, {k: r[k] for k in keys})
if os.environ.get("WITH_ORACLE"):
    import pyoracle
    pyoracle.load()
    wo = w.copy()
    ro, _ = pyoracle.solve(wo, o, num_threads=8)
    print("%-10s" % "oracle", {k: ro.get(k) for k in keys}, "pose diff vs gpu %.3e" % np.abs(wg.kf_pose - wo.kf_pose).max())
