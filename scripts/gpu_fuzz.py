"""Randomised parity sweep: GPU (single solves and one mixed batch) against the oracle on windows of random shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from limo_amd import ba, default_options, synth
import pyoracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
pyoracle.load()
ctx = ba.Context(0)
o = default_options()
TOL = 1e-4
ws, kws = [], []
for i in range(N):
    kw = dict(n_kf=int(rng.integers(3, 13)), n_lm=int(rng.choice([60, 120, 300, 700, 1500, 3000])),
              depth_prob=float(rng.choice([0.0, 0.02, 0.2, 0.45, 0.9])), ground_frac=float(rng.choice([0.0, 0.05, 0.2, 0.5])),
              outlier_frac=float(rng.choice([0.0, 0.05, 0.15])), stereo_baseline=float(rng.choice([0.0, 0.0, 0.54])),
              with_ground_plane=bool(rng.integers(0, 2)))
    kws.append(kw)
    ws.append(synth.make_window(10000 + i, **kw))
bad = 0
worst_c = worst_p = 0.0
singles = []
for i, (w, kw) in enumerate(zip(ws, kws)):
    wg, wo = w.copy(), w.copy()
    rg = ctx.solve(wg, o)
    ro, _ = pyoracle.solve(wo, o, num_threads=3)
    singles.append(wg)
    ec = abs(rg["final_cost"] - ro["final_cost"]) / max(1e-300, abs(ro["final_cost"]))
    ep = np.abs(wg.kf_pose[:, 4:] - wo.kf_pose[:, 4:]).max() / max(1e-12, np.abs(wo.kf_pose[:, 4:]).max())
    ok = (rg["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"] and rg["termination"] == ro["termination"] and ec <= TOL and ep <= TOL
          and np.array_equal(wg.kf_pose[0], w.kf_pose[0]))
    worst_c, worst_p = max(worst_c, ec), max(worst_p, ep)
    if not ok:
        bad += 1
        print("MISMATCH", i, kw, "gpu", rg["termination"], rg["n_trimmed_landmarks"], rg["final_cost"], rg["iterations_total"], "oracle", ro["termination"], ro["n_trimmed_landmarks"], ro["final_cost"], ro["iterations_total"], "ec %.2e ep %.2e" % (ec, ep), flush=True)
b = ba.Batch(ctx, [w.copy() for w in ws])
b.solve(o)
b.download()
nb = 0
for i, (ws_, wb) in enumerate(zip(singles, b.windows)):
    d = np.abs(ws_.kf_pose - wb.kf_pose).max()
    if d > 0.0:  # kernel variants are chosen per window: a batch reproduces the single solve bit for bit
        nb += 1
        print("BATCH != SINGLE", i, kws[i], d, flush=True)
print("fuzz: %d windows, %d oracle mismatches (worst rel cost %.2e, rel pose %.2e), %d batch/single differences" % (N, bad, worst_c, worst_p, nb))
sys.exit(1 if (bad or nb) else 0)
