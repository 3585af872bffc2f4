"""ctypes mirror of include/limo_hip.h and loader of the C-ABI shared library.

This is plumbing only: every structure below is a field-for-field copy of the C declaration it names.
The product path has NO CPU fallback: if `liblimo_hip.so` is missing, `load()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIMO_HIP_LIB") or os.path.join(_HERE, "lib", "liblimo_hip.so")  # LIMO_HIP_LIB: A/B another build of the SAME library

LIMO_FIX_POSE, LIMO_FIX_SCALE, LIMO_FIX_NONE = 0, 1, 2
LIMO_OK, LIMO_ERR_INVALID, LIMO_ERR_NOT_ENOUGH_KF, LIMO_ERR_RUNTIME, LIMO_ERR_NO_DEVICE = 0, -1, -2, -3, -4
LIMO_CONVERGENCE, LIMO_NO_CONVERGENCE, LIMO_FAILURE = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class BaWindow(C.Structure):  # struct limo_ba_window
    _fields_ = [
        ("n_kf", C.c_int32),
        ("n_cam", C.c_int32),
        ("n_lm", C.c_int32),
        ("n_obs", C.c_int32),
        ("kf_pose", c_double_p),
        ("kf_plane_dir", c_double_p),
        ("kf_plane_dist", c_double_p),
        ("kf_fixation", c_int32_p),
        ("cam", c_double_p),
        ("lm_pos", c_double_p),
        ("lm_weight", c_double_p),
        ("lm_is_ground", c_uint8_p),
        ("obs_kf", c_int32_p),
        ("obs_lm", c_int32_p),
        ("obs_cam", c_int32_p),
        ("obs_u", c_float_p),
        ("obs_v", c_float_p),
        ("obs_d", c_float_p),
    ]


class BaOptions(C.Structure):  # struct limo_ba_options
    _fields_ = [
        ("depth_thres", C.c_double),
        ("reprojection_thres", C.c_double),
        ("depth_quantile", C.c_double),
        ("reprojection_quantile", C.c_double),
        ("num_trim_rounds", C.c_int32),
        ("trim_solver_iterations", C.c_int32),
        ("min_landmarks_for_trimming", C.c_int32),
        ("minimum_number_residual_groups", C.c_int32),
        ("max_num_iterations", C.c_int32),
        ("max_solver_time_sec", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
    ]


class BaReport(C.Structure):  # struct limo_ba_report
    _fields_ = [
        ("termination", C.c_int32),
        ("num_solves", C.c_int32),
        ("iterations_total", C.c_int32),
        ("iterations_final", C.c_int32),
        ("successful_steps", C.c_int32),
        ("n_depth_blocks", C.c_int32),
        ("n_repr_blocks", C.c_int32),
        ("n_gp_blocks", C.c_int32),
        ("n_trimmed_landmarks", C.c_int32),
        ("num_linearizations", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("time_sec", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SpeedPrior(C.Structure):  # struct limo_speed_prior
    _fields_ = [
        ("speed_weight", C.c_double),
        ("dt_cur", C.c_double),
        ("vel_prev", C.c_double * 3),
        ("pose_before", C.c_double * 7),
    ]


class BaRow(C.Structure):  # struct limo_ba_row
    _fields_ = [
        ("kind", C.c_int32),
        ("sub", C.c_int32),
        ("kf", C.c_int32 * 2),
        ("lm", C.c_int32),
        ("fixed", C.c_int32),
        ("r", C.c_double),
        ("cost", C.c_double),
        ("jac_kf", (C.c_double * 10) * 2),
        ("jac_lm", C.c_double * 3),
    ]


ROW_GROUND_HEIGHT, ROW_SCALE, ROW_NORMAL_DIFF, ROW_DIST_DIFF, ROW_PLANE_MOTION, ROW_GLOBAL_NORMAL, ROW_SPEED = range(7)


def rows_as_dicts(rows, n):
    """limo_ba_row[n] -> list of dicts keyed the way the tests match rows: (kind, kf0, kf1, lm, sub)."""
    import numpy as np

    out = []
    for i in range(n):
        r = rows[i]
        out.append({"key": (r.kind, r.kf[0], r.kf[1], r.lm, r.sub), "fixed": r.fixed, "r": r.r, "cost": r.cost,
                    "jac_kf": np.array([list(r.jac_kf[0]), list(r.jac_kf[1])]), "jac_lm": np.array(list(r.jac_lm))})
    return out


class Ray(C.Structure):  # struct limo_ray
    _fields_ = [
        ("pose_cam_origin", C.c_double * 7),
        ("f", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("u", C.c_float),
        ("v", C.c_float),
        ("d", C.c_float),
        ("pad", C.c_float),
    ]


class DepthFrame(C.Structure):  # struct limo_depth_frame (pointers as integers: host arrays or device addresses)
    _fields_ = [
        ("cloud_xyzi", C.c_void_p),
        ("n_pts", C.c_size_t),
        ("feat_uv", C.c_void_p),
        ("n_feat", C.c_size_t),
        ("feat_is_ground", C.c_void_p),
        ("depth_out", C.c_void_p),
    ]


DEPTH_DEVICE_POINTERS = 1


class DepthParams(C.Structure):  # struct limo_depth_params
    _fields_ = [
        ("pixelarea_search_width", C.c_int32),
        ("pixelarea_search_height", C.c_int32),
        ("pixelarea_search_offset_x", C.c_int32),
        ("pixelarea_search_offset_y", C.c_int32),
        ("neighbors_count_min", C.c_int32),
        ("do_use_histogram_segmentation", C.c_int32),
        ("histogram_segmentation_bin_width", C.c_double),
        ("histogram_segmentation_min_pointcount", C.c_int32),
        ("treshold_depth_enabled", C.c_int32),
        ("treshold_depth_max", C.c_double),
        ("treshold_depth_min", C.c_double),
        ("treshold_depth_local_enabled", C.c_int32),
        ("treshold_depth_local_valuetype", C.c_int32),
        ("treshold_depth_local_value", C.c_double),
        ("do_use_cut_behind_camera", C.c_int32),
        ("do_use_triangle_size_maximation", C.c_int32),
        ("do_check_triangleplanar_condition", C.c_int32),
        ("triangleplanar_crossnorm_treshold", C.c_double),
        ("viewray_plane_orthoganality_treshold", C.c_double),
        ("do_use_ransac_plane", C.c_int32),
        ("ransac_plane_distance_treshold", C.c_double),
        ("ransac_plane_min_z", C.c_double),
        ("ransac_plane_max_z", C.c_double),
        ("ransac_plane_max_iterations", C.c_int32),
        ("ransac_plane_probability", C.c_double),
        ("ransac_plane_use_refinement", C.c_int32),
        ("ransac_plane_refinement_treshold", C.c_double),
        ("ransac_plane_point_distance_treshold", C.c_double),
        ("plane_estimator_use_mestimator", C.c_int32),
        ("ransac_seed", C.c_uint64),
    ]


# limo_exchange_fn: (send, recv, count, kind, user) - kind 0 all-gather, 1 sum over the ranks
EXCHANGE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_longlong, C.c_int, C.c_void_p)

ABI_VERSION = 5  # LIMO_ABI_VERSION of include/limo_hip.h

# every symbol include/limo_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "limo_abi_version",
    "limo_ctx_create",
    "limo_ctx_destroy",
    "limo_ctx_set_stream",
    "limo_last_error",
    "limo_host_alloc",
    "limo_host_free",
    "limo_ba_default_options",
    "limo_ba_solve",
    "limo_ba_batch_create",
    "limo_ba_batch_solve",
    "limo_ba_batch_reset",
    "limo_ba_batch_download",
    "limo_ba_batch_destroy",
    "limo_ba_batch_trimmed",
    "limo_ba_batch_kernel_stats",
    "limo_ba_batch_kernel_time",
    "limo_comm_unique_id",
    "limo_ctx_comm_init",
    "limo_ctx_comm_init_host",
    "limo_ba_solve_sharded",
    "limo_ctx_exchange_stats",
    "limo_ctx_coop_fallbacks",
    "limo_ba_evaluate",
    "limo_ba_evaluate_batch_time",
    "limo_ba_evaluate_rows",
    "limo_ba_adjust_pose_only",
    "limo_landmark_init",
    "limo_trim_quantile",
    "limo_depth_default_params",
    "limo_depth_estimate",
    "limo_depth_estimate_begin",
    "limo_depth_estimate_end",
    "limo_depth_estimate_batch",
    "limo_depth_last_ground_plane",
    "limo_depth_set_timing",
    "limo_depth_last_kernel_ms",
]

_lib = None


def load():
    """Load liblimo_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "limo_amd: %s not found - run `python -c 'import __graft_entry__ as g; g.build()'` first. "
            "There is no CPU fallback for the product path." % LIB_PATH
        )
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so); ours links /opt/rocm's.  A process that loads ours
    # first and torch afterwards aborts at interpreter exit ("double free or corruption"); the other order is clean - so
    # this Python plumbing (which uses torch for device tensors and torch.distributed anyway) fixes the order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.limo_abi_version.restype = C.c_int
    older = os.environ.get("LIMO_ALLOW_OLDER_ABI") == "1" and lib.limo_abi_version() < ABI_VERSION  # (A/B runs against an earlier round's build)
    if lib.limo_abi_version() != ABI_VERSION and not older:
        raise RuntimeError("limo_amd: %s has ABI version %d, this binding needs %d - rebuild it (__graft_entry__.build())" % (LIB_PATH, lib.limo_abi_version(), ABI_VERSION))
    lib.limo_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.limo_ctx_destroy.argtypes = [vp]
    lib.limo_ctx_destroy.restype = None
    lib.limo_ctx_set_stream.argtypes = [vp, vp]
    lib.limo_last_error.argtypes = [vp]
    lib.limo_last_error.restype = C.c_char_p
    lib.limo_ba_default_options.argtypes = [C.POINTER(BaOptions)]
    lib.limo_ba_default_options.restype = None
    lib.limo_ba_solve.argtypes = [vp, C.POINTER(BaWindow), C.POINTER(BaOptions), C.POINTER(BaReport)]
    lib.limo_ba_batch_create.argtypes = [vp, C.c_int32, C.POINTER(BaWindow), C.POINTER(vp)]
    lib.limo_ba_batch_solve.argtypes = [vp, C.POINTER(BaOptions)]
    lib.limo_ba_batch_reset.argtypes = [vp]
    lib.limo_ba_batch_download.argtypes = [vp, C.POINTER(BaWindow), C.POINTER(BaReport)]
    lib.limo_ba_batch_destroy.argtypes = [vp]
    lib.limo_ba_batch_destroy.restype = None
    lib.limo_ba_batch_trimmed.argtypes = [vp, C.c_int32, c_uint8_p]
    lib.limo_ba_batch_kernel_stats.argtypes = [vp, C.c_int, c_double_p, c_int64_p, c_double_p]
    lib.limo_ba_batch_kernel_time.argtypes = [vp, C.c_int, c_double_p, c_int64_p]
    lib.limo_comm_unique_id.argtypes = [C.c_char_p]
    lib.limo_ctx_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    if not older:
        lib.limo_ctx_comm_init_host.argtypes = [vp, EXCHANGE_FN, C.c_void_p, C.c_int, C.c_int]
    lib.limo_ba_solve_sharded.argtypes = [vp, C.POINTER(BaWindow), C.POINTER(BaOptions), C.c_int, C.POINTER(BaReport)]
    lib.limo_ba_evaluate_batch_time.argtypes = [vp, C.c_int32, C.POINTER(BaWindow), C.POINTER(BaOptions), C.c_int32, C.POINTER(C.c_double)]
    lib.limo_ba_evaluate.argtypes = [
        vp,
        C.POINTER(BaWindow),
        C.POINTER(BaOptions),
        C.c_int,
        c_double_p,
        c_double_p,
        c_double_p,
        c_double_p,
        c_uint8_p,
    ]
    lib.limo_ba_evaluate_rows.argtypes = [vp, C.POINTER(BaWindow), C.POINTER(SpeedPrior), C.c_int, C.POINTER(BaOptions), C.c_int32, C.POINTER(BaRow), c_int32_p]
    lib.limo_ba_adjust_pose_only.argtypes = [
        vp,
        C.POINTER(BaWindow),
        C.POINTER(SpeedPrior),
        C.POINTER(BaOptions),
        C.POINTER(BaReport),
    ]
    lib.limo_landmark_init.argtypes = [vp, C.c_int32, c_int32_p, C.POINTER(Ray), c_uint8_p, c_double_p, c_uint8_p]
    lib.limo_trim_quantile.argtypes = [C.c_int32, c_int64_p, c_double_p, C.c_double, c_int64_p]
    lib.limo_depth_default_params.argtypes = [C.POINTER(DepthParams)]
    lib.limo_depth_default_params.restype = None
    lib.limo_depth_estimate.argtypes = [
        vp,
        c_float_p,
        C.c_size_t,
        c_double_p,
        C.c_double,
        C.c_double,
        C.c_double,
        C.c_int32,
        C.c_int32,
        c_float_p,
        C.c_size_t,
        c_uint8_p,
        C.POINTER(DepthParams),
        c_float_p,
    ]
    lib.limo_depth_estimate_begin.argtypes = lib.limo_depth_estimate.argtypes[:-1]
    lib.limo_depth_estimate_end.argtypes = [vp, c_float_p, C.c_size_t]
    lib.limo_host_alloc.argtypes = [C.c_size_t]
    lib.limo_host_alloc.restype = C.c_void_p
    lib.limo_host_free.argtypes = [C.c_void_p]
    lib.limo_host_free.restype = None
    lib.limo_depth_estimate_batch.argtypes = [vp, C.c_int32, C.POINTER(DepthFrame), c_double_p, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                              C.POINTER(DepthParams), C.c_uint32]
    lib.limo_ctx_exchange_stats.argtypes = [vp, c_int64_p]
    lib.limo_ctx_coop_fallbacks.argtypes = [vp]
    lib.limo_ctx_coop_fallbacks.restype = C.c_int64
    lib.limo_depth_set_timing.argtypes = [vp, C.c_int32]
    lib.limo_depth_last_kernel_ms.argtypes = [vp, c_double_p]
    lib.limo_depth_last_ground_plane.argtypes = [vp, C.c_int32, c_double_p, c_int32_p]
    _lib = lib
    return lib
