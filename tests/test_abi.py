"""The C-ABI shared library loads and exports every symbol include/limo_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

from limo_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "limo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(limo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    declared = header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "liblimo_hip.so does not export %s" % name
    assert sorted(_ffi.ABI_SYMBOLS) == declared
    assert lib.limo_abi_version() == _ffi.ABI_VERSION


def test_struct_sizes_match_the_header_layout():
    # limo_ba_window: 4 int32 + 14 pointers; limo_ba_report: 10 int32 + 3 doubles
    assert C.sizeof(_ffi.BaWindow) == 16 + 14 * 8
    assert C.sizeof(_ffi.BaReport) == 40 + 24
    assert C.sizeof(_ffi.Ray) == 7 * 8 + 3 * 8 + 4 * 4
    assert C.sizeof(_ffi.SpeedPrior) == 8 * (2 + 3 + 7)


def test_default_options_are_the_reference_defaults():
    lib = _ffi.load()
    o = _ffi.BaOptions()
    lib.limo_ba_default_options(C.byref(o))
    # OutlierRejectionOptions, bundle_adjuster_keyframes.hpp:79-89
    assert (o.depth_thres, o.reprojection_thres, o.depth_quantile, o.reprojection_quantile, o.num_trim_rounds) == (0.16, 1.6, 0.95, 0.95, 1)
    # robust_solving.hpp:93-108, bundle_adjuster_keyframes.cpp:741-745,762
    assert (o.max_num_iterations, o.trim_solver_iterations, o.min_landmarks_for_trimming, o.minimum_number_residual_groups) == (100, 2, 100, 30)
    # Ceres 1.13 trust-region defaults
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_lm_diagonal, o.min_relative_decrease) == (1e4, 1e16, 1e-6, 1e-3)
    from limo_amd import default_options

    p = default_options()
    for name, _ in _ffi.BaOptions._fields_:
        assert getattr(o, name) == getattr(p, name), name


def test_no_context_without_a_gpu_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        return
    lib = _ffi.load()
    ptr = C.c_void_p()
    assert lib.limo_ctx_create(0, C.byref(ptr)) == _ffi.LIMO_ERR_NO_DEVICE


def test_trim_quantile_host_entry_point():
    import numpy as np

    lib = _ffi.load()
    ids = np.arange(110, dtype=np.int64)
    vals = np.r_[np.full(10, 5.0), np.linspace(0, 3.4, 100)]
    out = np.zeros(110, np.int64)
    n = lib.limo_trim_quantile(110, ids.ctypes.data_as(_ffi.c_int64_p), vals.ctypes.data_as(_ffi.c_double_p), 0.9, out.ctypes.data_as(_ffi.c_int64_p))
    # robust_optimization/test/robust_optimization.cpp:99-107: 110 residuals at quantile 0.9 -> 11 outliers
    assert n == 11
    assert set(range(10)) <= set(out[:n].tolist())
