"""Host-side container for one optimisation window (numpy-owned buffers behind a `limo_ba_window`).

Mirrors what BundleAdjusterKeyframes holds between "selection done" and the solver call
(reference: keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:641-643,695-698): active keyframes
(pose, local ground plane, fixation status), selected landmarks (position, weight, ground flag) and the
measurements that connect them.
"""
import ctypes as C

import numpy as np

from . import _ffi


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class Window:
    """Struct-of-arrays window.  All arrays are C-contiguous and owned by this object."""

    FIELDS = (
        ("kf_pose", np.float64),
        ("kf_plane_dir", np.float64),
        ("kf_plane_dist", np.float64),
        ("kf_fixation", np.int32),
        ("cam", np.float64),
        ("lm_pos", np.float64),
        ("lm_weight", np.float64),
        ("lm_is_ground", np.uint8),
        ("obs_kf", np.int32),
        ("obs_lm", np.int32),
        ("obs_cam", np.int32),
        ("obs_u", np.float32),
        ("obs_v", np.float32),
        ("obs_d", np.float32),
    )

    def __init__(self, **arrays):
        for name, dt in self.FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays[name], dtype=dt))
        self.meta = dict(arrays.get("meta", {}))
        self.kf_pose = self.kf_pose.reshape(-1, 7)
        self.kf_plane_dir = self.kf_plane_dir.reshape(-1, 3)
        self.cam = self.cam.reshape(-1, 10)
        self.lm_pos = self.lm_pos.reshape(-1, 3)
        self.validate()

    # sizes
    @property
    def n_kf(self):
        return self.kf_pose.shape[0]

    @property
    def n_cam(self):
        return self.cam.shape[0]

    @property
    def n_lm(self):
        return self.lm_pos.shape[0]

    @property
    def n_obs(self):
        return self.obs_kf.shape[0]

    def validate(self):
        K, N, M, Cn = self.n_kf, self.n_lm, self.n_obs, self.n_cam
        assert self.kf_plane_dir.shape == (K, 3) and self.kf_plane_dist.shape == (K,) and self.kf_fixation.shape == (K,)
        assert self.lm_weight.shape == (N,) and self.lm_is_ground.shape == (N,)
        for a in (self.obs_lm, self.obs_cam, self.obs_u, self.obs_v, self.obs_d):
            assert a.shape == (M,)
        if M:
            assert 0 <= self.obs_kf.min() and self.obs_kf.max() < K
            assert 0 <= self.obs_lm.min() and self.obs_lm.max() < N
            assert 0 <= self.obs_cam.min() and self.obs_cam.max() < Cn

    def copy(self):
        d = {name: getattr(self, name).copy() for name, _ in self.FIELDS}
        d["meta"] = dict(self.meta)
        return Window(**d)

    def as_struct(self):
        """limo_ba_window pointing into this object's buffers (keep `self` alive while it is used)."""
        s = _ffi.BaWindow()
        s.n_kf, s.n_cam, s.n_lm, s.n_obs = self.n_kf, self.n_cam, self.n_lm, self.n_obs
        s.kf_pose = _ptr(self.kf_pose, _ffi.c_double_p)
        s.kf_plane_dir = _ptr(self.kf_plane_dir, _ffi.c_double_p)
        s.kf_plane_dist = _ptr(self.kf_plane_dist, _ffi.c_double_p)
        s.kf_fixation = _ptr(self.kf_fixation, _ffi.c_int32_p)
        s.cam = _ptr(self.cam, _ffi.c_double_p)
        s.lm_pos = _ptr(self.lm_pos, _ffi.c_double_p)
        s.lm_weight = _ptr(self.lm_weight, _ffi.c_double_p)
        s.lm_is_ground = _ptr(self.lm_is_ground, _ffi.c_uint8_p)
        s.obs_kf = _ptr(self.obs_kf, _ffi.c_int32_p)
        s.obs_lm = _ptr(self.obs_lm, _ffi.c_int32_p)
        s.obs_cam = _ptr(self.obs_cam, _ffi.c_int32_p)
        s.obs_u = _ptr(self.obs_u, _ffi.c_float_p)
        s.obs_v = _ptr(self.obs_v, _ffi.c_float_p)
        s.obs_d = _ptr(self.obs_d, _ffi.c_float_p)
        return s

    # binary dump/load so oracle, CPU baseline and GPU can be fed identical bytes across processes
    def save(self, path):
        np.savez(path, **{name: getattr(self, name) for name, _ in self.FIELDS})

    @staticmethod
    def load(path):
        z = np.load(path)
        return Window(**{name: z[name] for name, _ in Window.FIELDS})


def struct_array(windows):
    """Contiguous C array of limo_ba_window for a list of Window objects."""
    arr = (_ffi.BaWindow * len(windows))()
    for i, w in enumerate(windows):
        arr[i] = w.as_struct()
    return arr


def default_options(**overrides):
    """limo_ba_options with the reference defaults (see include/limo_hip.h); overrides by field name."""
    o = _ffi.BaOptions()
    o.depth_thres = 0.16
    o.reprojection_thres = 1.6
    o.depth_quantile = 0.95
    o.reprojection_quantile = 0.95
    o.num_trim_rounds = 1
    o.trim_solver_iterations = 2
    o.min_landmarks_for_trimming = 100
    o.minimum_number_residual_groups = 30
    o.max_num_iterations = 100
    o.max_solver_time_sec = -1.0
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.min_relative_decrease = 1e-3
    o.max_num_consecutive_invalid_steps = 5
    o.jacobi_scaling = 1
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o
