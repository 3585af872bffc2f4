"""Generates tests/golden/oracle_windows.json: final poses / costs of the oracle on small seeded windows.

The reference cannot run here (no Ceres/Eigen), so these are ORACLE outputs, committed so that (a) the oracle cannot
drift silently and (b) the GPU tests have fixed targets that do not depend on rebuilding the oracle.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from limo_amd import default_options, synth  # noqa: E402

CASES = [
    {"seed": 1, "n_kf": 3, "n_lm": 200, "kw": {"depth_prob": 0.0, "ground_frac": 0.0, "with_ground_plane": False}},
    {"seed": 11, "n_kf": 4, "n_lm": 300},
    {"seed": 12, "n_kf": 5, "n_lm": 60},
    {"seed": 13, "n_kf": 7, "n_lm": 500},
]
out = {"generator": "tests/golden/make_golden.py", "cases": []}
for c in CASES:
    w = synth.make_window(c["seed"], n_kf=c["n_kf"], n_lm=c["n_lm"], **c.get("kw", {}))
    n_obs, n_kept = w.n_obs, w.n_lm
    rep, _ = pyoracle.solve(w, default_options())
    d = dict(c)
    d.update(n_obs=n_obs, n_lm_kept=n_kept, n_trimmed=rep["n_trimmed_landmarks"], initial_cost=rep["initial_cost"], final_cost=rep["final_cost"], kf_pose=w.kf_pose.tolist(), iterations=rep["iterations_total"])
    out["cases"].append(d)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_windows.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["cases"]), "cases")
