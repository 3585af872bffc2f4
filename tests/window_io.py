"""Reads the windows the C++ shim dumps with LIMO_KBA_DUMP (limo_amd/kba/bundle_adjuster_keyframes.cpp:dump_window_if_asked)
and stores / loads them as compressed .npz fixtures."""
import numpy as np

from limo_amd.window import Window


def read_dump(path):
    raw = open(path, "rb").read()
    n_kf, n_cam, n_lm, n_obs = np.frombuffer(raw, np.int32, 4)
    off = 16

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype, count, off).copy()
        off += a.nbytes
        return a

    d = dict(
        kf_pose=take(np.float64, 7 * n_kf), kf_plane_dir=take(np.float64, 3 * n_kf), kf_plane_dist=take(np.float64, n_kf),
        kf_fixation=take(np.int32, n_kf), cam=take(np.float64, 10 * n_cam), lm_pos=take(np.float64, 3 * n_lm),
        lm_weight=take(np.float64, n_lm), lm_is_ground=take(np.uint8, n_lm), obs_kf=take(np.int32, n_obs), obs_lm=take(np.int32, n_obs),
        obs_cam=take(np.int32, n_obs), obs_u=take(np.float32, n_obs), obs_v=take(np.float32, n_obs), obs_d=take(np.float32, n_obs),
    )
    assert off == len(raw)
    return Window(**d)


def save_npz(path, w):
    np.savez_compressed(path, **{name: getattr(w, name) for name, _ in Window.FIELDS})


def load_npz(path):
    z = np.load(path)
    return Window(**{name: z[name] for name, _ in Window.FIELDS})
