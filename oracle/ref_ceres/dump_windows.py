"""Writes the windows the dormant reference recipe (oracle/ref_ceres/Makefile, `make vectors`) solves with real Ceres, in the
LIMO_KBA_DUMP format, into oracle/_ref/windows/: the seeded windows of tests/golden/oracle_windows.json, the drive window
tests/golden/window_drive_frame1674.npz, C1 and C2.  Run from the repo root:  python oracle/ref_ceres/dump_windows.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from limo_amd import synth  # noqa: E402
import window_io  # noqa: E402


def write_dump(path, w):
    """Inverse of tests/window_io.py:read_dump."""
    with open(path, "wb") as f:
        np.array([w.n_kf, w.n_cam, w.n_lm, w.n_obs], np.int32).tofile(f)
        for name, dt in (("kf_pose", np.float64), ("kf_plane_dir", np.float64), ("kf_plane_dist", np.float64), ("kf_fixation", np.int32), ("cam", np.float64),
                         ("lm_pos", np.float64), ("lm_weight", np.float64), ("lm_is_ground", np.uint8), ("obs_kf", np.int32), ("obs_lm", np.int32),
                         ("obs_cam", np.int32), ("obs_u", np.float32), ("obs_v", np.float32), ("obs_d", np.float32)):
            np.ascontiguousarray(getattr(w, name), dt).tofile(f)


def windows():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_windows.json")))
    for c in gold["cases"]:
        yield "seed%d_kf%d_lm%d" % (c["seed"], c["n_kf"], c["n_lm"]), synth.make_window(c["seed"], n_kf=c["n_kf"], n_lm=c["n_lm"], **c.get("kw", {}))
    yield "drive_frame1674", window_io.load_npz(os.path.join(ROOT, "tests", "golden", "window_drive_frame1674.npz"))
    yield "c1", synth.config_c1()
    yield "c2", synth.config_c2()


if __name__ == "__main__":
    out = os.path.join(ROOT, "oracle", "_ref", "windows")
    os.makedirs(out, exist_ok=True)
    for name, w in windows():
        write_dump(os.path.join(out, name + ".bin"), w)
        print(name, w.n_kf, w.n_lm, w.n_obs)
