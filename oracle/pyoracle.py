"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from limo_amd import _ffi  # noqa: E402  (struct layouts only)

LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    dp, ip, u8p = _ffi.c_double_p, _ffi.c_int32_p, _ffi.c_uint8_p
    lib.oracle_ba_default_options.argtypes = [C.POINTER(_ffi.BaOptions)]
    lib.oracle_ba_default_options.restype = None
    lib.oracle_ba_solve.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.POINTER(_ffi.BaReport), C.c_int, C.c_int, dp]
    lib.oracle_last_trimmed.argtypes = [ip, C.c_int]
    lib.oracle_ba_adjust_pose_only.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.SpeedPrior), C.POINTER(_ffi.BaOptions), C.POINTER(_ffi.BaReport), C.c_int]
    lib.oracle_ba_evaluate.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), C.c_int, dp, dp, dp, dp, u8p]
    lib.oracle_ba_evaluate_rows.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.SpeedPrior), C.c_int, C.POINTER(_ffi.BaOptions), C.c_int32, C.POINTER(_ffi.BaRow), ip]
    lib.oracle_ba_problem_cost.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), dp, ip]
    lib.oracle_trim_quantile.argtypes = [C.c_int32, _ffi.c_int64_p, dp, C.c_double, _ffi.c_int64_p]
    lib.oracle_trim_fix.argtypes = [C.c_int32, _ffi.c_int64_p, dp, C.c_double, _ffi.c_int64_p]
    lib.oracle_landmark_init.argtypes = [C.c_int32, ip, C.POINTER(_ffi.Ray), u8p, dp, u8p]
    lib.oracle_functor.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp]
    lib.oracle_loss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp]
    lib.oracle_loss.restype = None
    lib.oracle_plus.argtypes = [C.c_int, dp, dp, dp, dp]
    lib.oracle_plus.restype = None
    lib.oracle_robust_test_solve_trimmed.argtypes = [C.c_int, C.c_int, C.c_double, ip, C.c_int, C.c_double]
    lib.oracle_robust_test_solve_trimmed.restype = C.c_double
    lib.oracle_num_procs.restype = C.c_int
    lib.oracle_functor_jacobian.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp]
    lib.oracle_ba_first_step.argtypes = [C.POINTER(_ffi.BaWindow), C.POINTER(_ffi.BaOptions), ip, dp, dp, dp, dp]
    _lib = lib
    return lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(_ffi.c_double_p)


def solve(window, opts, num_threads=1, num_linear_solver_threads=1):
    """Run the restated solve() on `window` IN PLACE.  Returns (report dict, phase_times[4])."""
    lib = load()
    s = window.as_struct()
    rep = _ffi.BaReport()
    pt = np.zeros(4)
    rc = lib.oracle_ba_solve(C.byref(s), C.byref(opts), C.byref(rep), num_threads, num_linear_solver_threads, _dp(pt))
    if rc != 0:
        raise RuntimeError("oracle_ba_solve rc=%d" % rc)
    return rep.as_dict(), pt


def last_trimmed():
    """Landmark indices (caller's order) removed by the trimming rounds of the last solve on this thread."""
    lib = load()
    n = lib.oracle_last_trimmed(None, 0)
    out = np.zeros(max(1, n), np.int32)
    lib.oracle_last_trimmed(out.ctypes.data_as(_ffi.c_int32_p), n)
    return np.sort(out[:n])


def adjust_pose_only(window, prior, opts, num_threads=1):
    lib = load()
    s = window.as_struct()
    rep = _ffi.BaReport()
    rc = lib.oracle_ba_adjust_pose_only(C.byref(s), None if prior is None else C.byref(prior), C.byref(opts), C.byref(rep), num_threads)
    if rc != 0:
        raise RuntimeError("oracle_ba_adjust_pose_only rc=%d" % rc)
    return rep.as_dict()


def evaluate(window, opts, apply_loss=True):
    lib = load()
    s = window.as_struct()
    M = window.n_obs
    cost = np.zeros(1)
    res = np.zeros((M, 3))
    jp = np.zeros((M, 3, 6))
    jl = np.zeros((M, 3, 3))
    valid = np.zeros(M, np.uint8)
    rc = lib.oracle_ba_evaluate(C.byref(s), C.byref(opts), int(apply_loss), _dp(cost), _dp(res), _dp(jp), _dp(jl), valid.ctypes.data_as(_ffi.c_uint8_p))
    if rc != 0:
        raise RuntimeError("oracle_ba_evaluate rc=%d" % rc)
    return float(cost[0]), res, jp, jl, valid


def evaluate_rows(window, opts, pose_only=False, prior=None):
    """The non-observation residual rows of the solve() / adjustPoseOnly problem with dual-number tangent Jacobians."""
    lib = load()
    s = window.as_struct()
    n = C.c_int32(0)
    cap = 64 + window.n_lm + 8 * window.n_kf
    rows = (_ffi.BaRow * cap)()
    rc = lib.oracle_ba_evaluate_rows(C.byref(s), None if prior is None else C.byref(prior), int(pose_only), C.byref(opts), cap, rows, C.byref(n))
    if rc != 0:
        raise RuntimeError("oracle_ba_evaluate_rows rc=%d" % rc)
    return _ffi.rows_as_dicts(rows, n.value)


def problem_cost(window, opts):
    lib = load()
    s = window.as_struct()
    cost = np.zeros(1)
    counts = np.zeros(3, np.int32)
    lib.oracle_ba_problem_cost(C.byref(s), C.byref(opts), _dp(cost), counts.ctypes.data_as(_ffi.c_int32_p))
    return float(cost[0]), counts


def functor(kind, consts, *params, nres=3):
    lib = load()
    consts = np.ascontiguousarray(consts if consts is not None else [0.0], np.float64)
    ps = [np.ascontiguousarray(p, np.float64) for p in params] + [None] * (4 - len(params))
    out = np.zeros(3)
    ok = lib.oracle_functor(kind, _dp(consts), _dp(ps[0]), _dp(ps[1]), _dp(ps[2]), _dp(ps[3]), _dp(out))
    return ok, out[:nres]


def functor_jacobian(kind, consts, *params):
    """(residuals, [ambient Jacobian per parameter block]) of functor kinds 0, 1, 3 by the oracle's dual numbers."""
    lib = load()
    consts = np.ascontiguousarray(consts if consts is not None else [0.0], np.float64)
    ps = [np.ascontiguousarray(p, np.float64) for p in params] + [None] * (4 - len(params))
    res = np.zeros(3)
    jac = np.zeros(64)
    n = lib.oracle_functor_jacobian(kind, _dp(consts), _dp(ps[0]), _dp(ps[1]), _dp(ps[2]), _dp(ps[3]), _dp(res), _dp(jac))
    if n <= 0:
        raise RuntimeError("oracle_functor_jacobian rc=%d" % n)
    out, off = [], 0
    for p in params:
        out.append(jac[off:off + n * len(p)].reshape(n, len(p)).copy())
        off += n * len(p)
    return res[:n], out


def first_step(window, opts):
    """The linear least-squares problem of the first LM step of solve() and the Schur-based solution: dict with J
    (column-scaled, landmarks first), r, D, y, num_e; None when the solve takes no step."""
    lib = load()
    s = window.as_struct()
    sizes = np.zeros(3, np.int32)
    rc = lib.oracle_ba_first_step(C.byref(s), C.byref(opts), sizes.ctypes.data_as(_ffi.c_int32_p), None, None, None, None)
    if rc == 1:
        return None  # the solve took no step (evaluation failed at x0, or it stopped at iteration zero)
    if rc != 0:
        raise RuntimeError("oracle_ba_first_step rc=%d" % rc)
    m, n = int(sizes[0]), int(sizes[1])
    J, r, D, y = np.zeros((m, n)), np.zeros(m), np.zeros(n), np.zeros(n)
    rc = lib.oracle_ba_first_step(C.byref(s), C.byref(opts), sizes.ctypes.data_as(_ffi.c_int32_p), _dp(J), _dp(r), _dp(D), _dp(y))
    if rc != 0:
        raise RuntimeError("oracle_ba_first_step rc=%d" % rc)
    return {"J": J, "r": r, "D": D, "y": y, "num_e": int(sizes[2])}


def loss(kind, a, weight, s):
    lib = load()
    rho = np.zeros(3)
    lib.oracle_loss(kind, a, weight, s, _dp(rho))
    return rho


def plus(kind, x, delta):
    lib = load()
    x = np.ascontiguousarray(x, np.float64)
    delta = np.ascontiguousarray(delta, np.float64)
    n = 7 if kind == 0 else 3
    l = 6 if kind == 0 else 3
    out = np.zeros(n)
    jac = np.zeros((n, l))
    lib.oracle_plus(kind, _dp(x), _dp(delta), _dp(out), _dp(jac))
    return out, jac


def trim_quantile(ids, values, q):
    lib = load()
    ids = np.ascontiguousarray(ids, np.int64)
    values = np.ascontiguousarray(values, np.float64)
    out = np.zeros(len(ids), np.int64)
    n = lib.oracle_trim_quantile(len(ids), ids.ctypes.data_as(_ffi.c_int64_p), _dp(values), q, out.ctypes.data_as(_ffi.c_int64_p))
    return out[:n]


def trim_fix(ids, values, thres):
    lib = load()
    ids = np.ascontiguousarray(ids, np.int64)
    values = np.ascontiguousarray(values, np.float64)
    out = np.zeros(len(ids), np.int64)
    n = lib.oracle_trim_fix(len(ids), ids.ctypes.data_as(_ffi.c_int64_p), _dp(values), thres, out.ctypes.data_as(_ffi.c_int64_p))
    return out[:n]


def landmark_init(ray_off, rays, use_depth):
    lib = load()
    n = len(ray_off) - 1
    ray_off = np.ascontiguousarray(ray_off, np.int32)
    use_depth = np.ascontiguousarray(use_depth, np.uint8)
    pos = np.zeros((n, 3))
    ok = np.zeros(n, np.uint8)
    lib.oracle_landmark_init(n, ray_off.ctypes.data_as(_ffi.c_int32_p), rays, use_depth.ctypes.data_as(_ffi.c_uint8_p), _dp(pos), ok.ctypes.data_as(_ffi.c_uint8_p))
    return pos, ok


def depth_default_params():
    lib = load()
    p = _ffi.DepthParams()
    lib.oracle_depth_default_params.argtypes = [C.POINTER(_ffi.DepthParams)]
    lib.oracle_depth_default_params.restype = None
    lib.oracle_depth_default_params(C.byref(p))
    return p


def depth_estimate(frame, params=None, use_ground_labels=True):
    lib = load()
    lib.oracle_depth_estimate.argtypes = [_ffi.c_float_p, C.c_size_t, _ffi.c_double_p, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _ffi.c_float_p, C.c_size_t, _ffi.c_uint8_p, C.POINTER(_ffi.DepthParams), _ffi.c_float_p]
    p = params if params is not None else depth_default_params()
    cloud = np.ascontiguousarray(frame["cloud"], np.float32)
    uv = np.ascontiguousarray(frame["uv"], np.float32)
    T = np.ascontiguousarray(frame["T_cam_lidar"], np.float64)
    g = np.ascontiguousarray(frame["is_ground"], np.uint8) if use_ground_labels else None
    out = np.zeros(uv.shape[0], np.float32)
    rc = lib.oracle_depth_estimate(cloud.ctypes.data_as(_ffi.c_float_p), cloud.shape[0], _dp(T), frame["f"], frame["cx"], frame["cy"], frame["w"], frame["h"], uv.ctypes.data_as(_ffi.c_float_p), uv.shape[0], None if g is None else g.ctypes.data_as(_ffi.c_uint8_p), C.byref(p), out.ctypes.data_as(_ffi.c_float_p))
    if rc != 0:
        raise RuntimeError("oracle_depth_estimate rc=%d" % rc)
    return out


def ground_plane(frame, params=None):
    lib = load()
    lib.oracle_ground_plane.argtypes = [_ffi.c_float_p, C.c_size_t, _ffi.c_double_p, C.POINTER(_ffi.DepthParams), _ffi.c_double_p]
    p = params if params is not None else depth_default_params()
    cloud = np.ascontiguousarray(frame["cloud"], np.float32)
    T = np.ascontiguousarray(frame["T_cam_lidar"], np.float64)
    pl = np.zeros(4)
    n = lib.oracle_ground_plane(cloud.ctypes.data_as(_ffi.c_float_p), cloud.shape[0], _dp(T), C.byref(p), _dp(pl))
    return n, pl
