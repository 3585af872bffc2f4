"""The device forms of kba_math.hpp's reciprocal helpers against IEEE arithmetic (VERDICT r05 item 1 ii).

Since round 5 every device path takes 1 / z (ReprojectionErrorWithQuaternions, cost_functors_ceres.hpp:116-152), the Cauchy
corrector's 1 / sqrt(1 + s / a^2) (bundle_adjuster_keyframes.cpp:589-591,616-618 through Ceres 1.13 corrector.cc) and the pivots
of the 3 x 3 landmark factor (SchurEliminator) from v_rcp_f64 / v_rsq_f64 + Newton steps - not correctly rounded.  The CPU-tier
emulation keeps IEEE division and square root, so only the GPU sees these; tests/cpp/math_probe.hip runs them on arrays.
Bars: <= 2 ulp against the correctly rounded value over the operand ranges the call sites can see (camera depths from the
functor's failure band |z| = 0.01 up, Cauchy sums >= 1, pivots from ~1e-22 (radius 1e16) to 1e16), exact `ok` flags.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def probe():
    import emu_ffi

    lib = C.CDLL(emu_ffi.MATH_PROBE_LIB)  # (built by __graft_entry__.build(); fails loudly when missing)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.probe_unary.argtypes = [C.c_int, dp, dp, C.c_int]
    lib.probe_chol3.argtypes = [dp, dp, ip, C.c_int]
    lib.probe_view_xy.argtypes = [dp, dp, dp, C.c_int]
    return lib


def _ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _ulps(got, exact_ld):
    """|got - exact| in units of the spacing of doubles at the exact value (exact in long double)."""
    ref = exact_ld.astype(np.float64)
    return np.abs((got.astype(np.longdouble) - exact_ld) / np.spacing(np.abs(ref)).astype(np.longdouble)).astype(np.float64)


def _operands(rng):
    x = np.concatenate([
        10.0 ** rng.uniform(-22, 16, 200000),            # log-spaced over everything a pivot or a depth can be
        rng.uniform(0.01, 0.0101, 20000),                # the edge of the functor's failure band
        rng.uniform(1.0, 1.0 + 1e-6, 20000),             # Cauchy sums of tiny residuals
        rng.uniform(1.0, 4.0, 50000),                    # one binade pair: every mantissa pattern class
        np.array([0.01, 1.0, 2.0, 4.0, 1e-22, 1e16, 1e300, 0.5, 3.0, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52]),
    ])
    return np.ascontiguousarray(x)


def test_rcp_nr_is_within_two_ulp_of_ieee_division(probe):
    rng = np.random.default_rng(1)
    x = _operands(rng)
    x = x[x < 1e200]
    x = np.ascontiguousarray(np.concatenate([x, -x]))  # both signs: camera depths behind the camera
    y = np.empty_like(x)
    assert probe.probe_unary(0, _ptr(x), _ptr(y), x.size) == 0
    u = _ulps(y, np.longdouble(1.0) / x.astype(np.longdouble))
    assert np.isfinite(y).all() and u.max() <= 2.0, (u.max(), x[u.argmax()])
    print("rcp_nr: %d operands, max %.3f ulp, %.4f %% not correctly rounded" % (x.size, u.max(), 100.0 * (y != 1.0 / x).mean()))


def test_rsqrt_nr_is_within_two_ulp_of_ieee(probe):
    rng = np.random.default_rng(2)
    x = _operands(rng)
    y = np.empty_like(x)
    assert probe.probe_unary(1, _ptr(x), _ptr(y), x.size) == 0
    u = _ulps(y, np.longdouble(1.0) / np.sqrt(x.astype(np.longdouble)))
    assert np.isfinite(y).all() and u.max() <= 2.0, (u.max(), x[u.argmax()])
    print("rsqrt_nr: %d operands, max %.3f ulp" % (x.size, u.max()))


def _chol3_ref(A):
    """The statements of kba_math.hpp:chol3_inv in long double (correctly rounded operations at 64-bit mantissa)."""
    A = A.astype(np.longdouble)
    a00, a01, a02, a11, a12, a22 = (A[:, k] for k in range(6))
    one = np.longdouble(1.0)
    ok0 = a00 > 0
    i00 = one / np.sqrt(np.where(ok0, a00, one))
    l10, l20 = a01 * i00, a02 * i00
    d1 = a11 - l10 * l10
    ok1 = d1 > 0
    i11 = one / np.sqrt(np.where(ok1, d1, one))
    l21 = (a12 - l20 * l10) * i11
    d2 = a22 - l20 * l20 - l21 * l21
    ok2 = d2 > 0
    i22 = one / np.sqrt(np.where(ok2, d2, one))
    li1 = -l10 * i00 * i11
    Li = np.stack([i00, li1, i11, -(l20 * i00 + l21 * li1) * i22, -l21 * i11 * i22, i22], axis=1)
    return Li, ok0 & ok1 & ok2, d1, d2


def test_chol3_inv_pivots_and_flags(probe):
    """Damped landmark blocks (V' + D^2): SPD matrices built from a known factor L L^T at scales 1e-22 .. 1e16 (what
    min_lm_diagonal 1e-6 / radius 1e16 and a close landmark seen by twenty views span), plus matrices that are NOT positive
    definite at each of the three pivots."""
    rng = np.random.default_rng(3)
    n = 60000
    scale = 10.0 ** rng.uniform(-11, 8, (n, 1))  # entries of A = (scale L)(scale L)^T: 1e-22 .. 1e16
    L = np.zeros((n, 3, 3))
    L[:, 0, 0], L[:, 1, 1], L[:, 2, 2] = rng.uniform(0.3, 1.0, n), rng.uniform(0.3, 1.0, n), rng.uniform(0.3, 1.0, n)
    L[:, 1, 0], L[:, 2, 0], L[:, 2, 1] = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    L *= scale[:, :, None]
    M = L @ np.transpose(L, (0, 2, 1))
    A = np.ascontiguousarray(np.stack([M[:, 0, 0], M[:, 0, 1], M[:, 0, 2], M[:, 1, 1], M[:, 1, 2], M[:, 2, 2]], axis=1))
    Li = np.empty_like(A)
    ok = np.empty(n, dtype=np.int32)
    assert probe.probe_chol3(_ptr(A), _ptr(Li), _ptr(ok, C.c_int), n) == 0
    ref, ok_ref, d1, d2 = _chol3_ref(A)
    # pivots that survive their subtraction with at least a thousandth of the diagonal: the factor is determined to ~1e-12
    well = ok_ref & (np.abs(d1) > 1e-3 * np.abs(A[:, 3])) & (np.abs(d2) > 1e-3 * np.abs(A[:, 5]))
    assert well.mean() > 0.5
    assert (ok[well] == 1).all()
    rel = np.abs((Li.astype(np.longdouble) - ref) / np.maximum(np.abs(ref), np.abs(ref).max(axis=1, keepdims=True) * 1e-3)).astype(np.float64)
    assert rel[well].max() <= 1e-11, rel[well].max()
    # the diagonal of the inverse factor (no cancellation beyond the pivot's own): a few ulp of the pivot's relative error
    print("chol3_inv: %d blocks, %d well-conditioned, max rel error of Bt entries %.2e" % (n, int(well.sum()), rel[well].max()))
    # not positive definite: each pivot in turn is <= 0 -> ok = false, everything stays finite
    bad = np.array([
        [0.0, 0.1, 0.1, 1.0, 0.1, 1.0],      # a00 = 0
        [-1.0, 0.1, 0.1, 1.0, 0.1, 1.0],     # a00 < 0
        [1.0, 2.0, 0.0, 1.0, 0.0, 1.0],      # d1 = 1 - 4 < 0
        [1.0, 1.0, 0.0, 1.0, 0.0, 1.0],      # d1 = 0
        [1.0, 0.0, 2.0, 1.0, 0.0, 1.0],      # d2 = 1 - 4 < 0
        [1.0, 0.0, 0.0, 1.0, 1.0, 1.0],      # d2 = 0
        [1.0, 0.5, 0.2, 2.0, 0.3, 3.0],      # positive definite (control)
    ])
    Lb = np.empty_like(bad)
    okb = np.empty(len(bad), dtype=np.int32)
    assert probe.probe_chol3(_ptr(bad), _ptr(Lb), _ptr(okb, C.c_int), len(bad)) == 0
    assert okb.tolist() == [0, 0, 0, 0, 0, 0, 1] and np.isfinite(Lb).all()


def test_view_xy_at_the_failure_band(probe):
    """xn = x / z, yn = y / z, 1 / z with z on both sides of |z| = 0.01 (cost_functors_ceres.hpp:78-83: the functor fails
    inside the band): the flag is exact, the quotients are within 4.5 ulp of the exact quotients, the band stays finite."""
    rng = np.random.default_rng(4)
    n = 40000
    vl = np.zeros((n, 12))
    vl[:, 0] = vl[:, 4] = vl[:, 8] = 1.0  # H = I, h0 = 0: the camera point is the landmark itself (exactly)
    p = np.empty((n, 3))
    p[:, 0], p[:, 1] = rng.uniform(-50, 50, n), rng.uniform(-10, 10, n)
    z = np.concatenate([rng.uniform(0.0099, 0.0101, n // 2), 10.0 ** rng.uniform(-2, 2.5, n - n // 2)])
    z[: n // 8] = 0.01
    z[n // 8 : n // 4] = np.nextafter(0.01, 0.0)
    p[:, 2] = z * np.where(rng.random(n) < 0.25, -1.0, 1.0)
    out = np.empty((n, 4))
    vl, p = np.ascontiguousarray(vl), np.ascontiguousarray(p)
    assert probe.probe_view_xy(_ptr(vl), _ptr(p), _ptr(out), n) == 0
    ok_ref = np.abs(p[:, 2]) >= 0.01
    assert np.array_equal(out[:, 3] == 1.0, ok_ref) and np.isfinite(out).all()
    zl = p[ok_ref, 2].astype(np.longdouble)
    for col, num in ((0, p[ok_ref, 0]), (1, p[ok_ref, 1])):
        u = _ulps(out[ok_ref, col], num.astype(np.longdouble) / zl)
        assert u.max() <= 4.5, u.max()  # (2 ulp of 1 / z can be 4 ulp of the product when the mantissas sit at opposite ends of their binades)
    assert _ulps(out[ok_ref, 2], np.longdouble(1.0) / zl).max() <= 2.0
    # inside the band the depth is replaced by 1: the coordinates are the numerators themselves
    assert np.array_equal(out[~ok_ref, 0], p[~ok_ref, 0]) and np.array_equal(out[~ok_ref, 2], np.ones((~ok_ref).sum()))
