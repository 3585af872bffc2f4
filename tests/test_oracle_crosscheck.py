"""Algorithm-independent checks of the BA oracle (SURVEY.md §8c "oracle design"): the oracle is a restatement of Ceres
1.13 written by the same hands as the GPU path, so this file looks at it from directions that share nothing with it.

  1. cost: tests/ba_numpy.py (vectorised numpy statement of SURVEY Appendix B) gives the oracle's cost at the start and at
     the solution, trimmed landmarks removed - functors, losses, ground-plane wiring, regulariser weights, trimming;
  2. minimum: scipy.optimize.least_squares (trust-region-reflective, dense SVD steps, complex-step Jacobian of the numpy
     residual vector) started at the oracle's solution stays there - the point the restated LM + Schur loop
     (robust_solving.cpp:169,174,239 -> ceres::Solve) stops at IS the minimum of the robustified cost;
  3. Jacobians: sympy derivatives of the reprojection / depth / ground-height residuals (cost_functors_ceres.hpp:91-155,
     193-212,358-385) against the oracle's dual numbers and against Problem::Evaluate's local Jacobians;
  4. properties (hypothesis): the residual vector does not change under a rigid change of the origin frame (gauge); the
     step of the Schur-complement linear solver equals the dense least-squares solution of the same damped problem.
"""
import numpy as np
import pytest
import scipy.optimize
from hypothesis import assume, given, settings
from hypothesis import strategies as st

import ba_numpy
from limo_amd import default_options, synth

TIGHT = dict(function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_num_iterations=300)


def solved(oracle, w0, **opts):
    """(numpy problem wired at the start point and moved to the oracle's solution, oracle report)"""
    o = default_options(**opts)
    ws = w0.copy()
    rep, _ = oracle.solve(ws, o)
    P = ba_numpy.Problem(w0.copy(), o)
    P.remove(oracle.last_trimmed())
    for name in ("kf_pose", "kf_plane_dir", "kf_plane_dist", "lm_pos"):
        getattr(P.w, name)[:] = getattr(ws, name)
    return P, rep, ws


CASES = {
    "c1": lambda: synth.config_c1(),
    "kf3_lm80": lambda: synth.make_window(11, n_kf=3, n_lm=80, depth_prob=0.3),
    "ground_kf4_lm400": lambda: synth.make_window(7, n_kf=4, n_lm=400),
    "c2": lambda: synth.config_c2(),
}


@pytest.mark.parametrize("case", list(CASES))
def test_numpy_statement_gives_the_oracle_cost(oracle, case):
    w = CASES[case]()
    o = default_options()
    c0, counts = oracle.problem_cost(w, o)
    P = ba_numpy.Problem(w.copy(), o)
    assert (P.n_depth, P.n_gp) == (counts[0], counts[2])
    assert abs(P.cost() - c0) <= 1e-12 * c0
    P, rep, _ = solved(oracle, w)
    assert abs(P.cost() - rep["final_cost"]) <= 1e-12 * rep["final_cost"]
    if case == "c2":
        assert rep["n_trimmed_landmarks"] > 50  # the trimming branch is part of what was compared


def polish(P):
    res = scipy.optimize.least_squares(P.residuals, np.zeros(P.n_free), jac="cs", method="trf", tr_solver="exact", x_scale="jac",
                                       ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=50)
    return res


@pytest.mark.parametrize("case", ["kf3_lm80", "c1", "trimmed_kf5_lm300"])
def test_scipy_confirms_the_minimum(oracle, case):
    """No-trimming cases and the post-trim problem of a C2-shaped window (5 keyframes, depth + ground plane, 300 landmarks
    so that scipy's dense trust-region solver applies; trimming is active from 100 landmarks on)."""
    w = synth.make_window(2, n_kf=5, n_lm=300) if case == "trimmed_kf5_lm300" else CASES[case]()
    # (a) the oracle run to convergence: scipy cannot lower the cost by more than 1e-8 and does not move the poses
    P, rep, _ = solved(oracle, w, **TIGHT)
    if case == "trimmed_kf5_lm300":
        assert rep["n_trimmed_landmarks"] >= 10
    c_oracle = P.cost()
    res = polish(P)
    assert res.cost <= c_oracle * (1 + 1e-12)
    assert (c_oracle - res.cost) <= 1e-8 * c_oracle
    base = P.w.kf_pose.copy()
    P.apply(res.x)
    scale = np.abs(base[:, 4:]).max()
    assert np.abs(P.w.kf_pose[:, 4:] - base[:, 4:]).max() <= 1e-7 * scale
    # (b) with the reference's tolerances (function_tolerance 1e-6) the stop is within that tolerance of the same minimum
    P2, rep2, ws2 = solved(oracle, w)
    assert 0 <= P2.cost() - res.cost <= 2e-6 * res.cost
    assert np.abs(ws2.kf_pose[:, 4:] - P.w.kf_pose[:, 4:]).max() <= 1e-4 * scale


def polish_sparse(P):
    """the same for windows too large for a dense Jacobian: complex-step differences grouped by the sparsity pattern,
    LSMR trust-region steps"""
    return scipy.optimize.least_squares(P.residuals, np.zeros(P.n_free), jac="cs", jac_sparsity=P.sparsity(), method="trf", tr_solver="lsmr",
                                        x_scale="jac", ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=40)


def test_sparsity_pattern_covers_the_dense_jacobian():
    w = synth.make_window(7, n_kf=4, n_lm=60)
    P = ba_numpy.Problem(w, default_options())
    rng = np.random.default_rng(0)
    x0 = 1e-3 * rng.normal(size=P.n_free)
    J = np.array([P.residuals(x0 + 1e-30j * np.eye(P.n_free)[i]).imag / 1e-30 for i in range(P.n_free)]).T
    S = P.sparsity().toarray()
    assert S.shape == J.shape and not (np.abs(J) > 0)[S == 0].any()


PLATEAU = {  # tests/fuzz_common.py:random_windows(n, seed)[idx] without generating the windows before it
    (77, 21): dict(n_kf=6, n_lm=3000, depth_prob=0.9, ground_frac=0.05, outlier_frac=0.15, stereo_baseline=0.54, with_ground_plane=True),
    (123, 115): dict(n_kf=6, n_lm=3000, depth_prob=0.2, ground_frac=0.05, outlier_frac=0.15, stereo_baseline=0.0, with_ground_plane=True),
}


@pytest.mark.parametrize("key", list(PLATEAU))
def test_plateau_windows_polish_into_their_own_basins(oracle, key):
    """The two fuzz windows whose final cost is not determined to 1e-4 by their input (tests/fuzz_common.py: seed 77 #21,
    seed 123 #115): the oracle's end point and the end point of the oracle on a 1-ulp-perturbed input are each polished by
    scipy.  Recorded (printed) per end point: the cost the oracle stopped at and the cost of the minimum below it.  What is
    asserted: scipy only ever goes down, and the poses of the two end points agree far inside 1e-4 before and after - the
    cost spread between them is the outliers' Cauchy basins, not the poses the pipeline consumes."""
    seed, idx = key
    w = synth.make_window(10000 + idx, **PLATEAU[key])
    ends = []
    for perturb in (False, True):
        wi = w.copy()
        if perturb:
            wi.lm_pos[0, 0] = np.nextafter(wi.lm_pos[0, 0], np.inf)
        P, rep, ws = solved(oracle, wi)
        res = polish_sparse(P)
        P.apply(res.x)
        ends.append((rep["final_cost"], res.cost, ws.kf_pose[:, 4:].copy(), P.w.kf_pose[:, 4:].copy()))
    (c0, p0, t0, u0), (c1, p1, t1, u1) = ends
    print("seed %d #%d: oracle %.4f -> polished %.4f | 1-ulp input: oracle %.4f -> polished %.4f" % (seed, idx, c0, p0, c1, p1))
    assert p0 <= c0 * (1 + 1e-9) and p1 <= c1 * (1 + 1e-9)
    scale = np.abs(t0).max()
    assert np.abs(t0 - t1).max() <= 1e-4 * scale and np.abs(u0 - u1).max() <= 1e-4 * scale
    assert np.abs(u0 - t0).max() <= 1e-4 * scale  # the polish does not move the poses either


# ------------------------------------------------------------------------------------------------------------------ sympy
def _sym_setup():
    import sympy as sp

    q = sp.symbols("qw qx qy qz tx ty tz", real=True)
    p = sp.symbols("px py pz", real=True)
    c = sp.symbols("cw cx_ cy_ cz ctx cty ctz", real=True)

    def R(w, x, y, z):
        return sp.Matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    X = R(*q[:4]) * sp.Matrix(p) + sp.Matrix(q[4:])
    Y = R(*c[:4]) * X + sp.Matrix(c[4:])
    return sp, q, p, c, X, Y


def test_sympy_jacobians_of_reprojection_depth_and_ground_height(oracle):
    sp, q, p, c, X, Y = _sym_setup()
    f, cx, cy, u, v, d = sp.symbols("f cx cy u v d", real=True)
    n = sp.symbols("nx ny nz", real=True)
    h = sp.Symbol("h", real=True)
    r_repr = sp.Matrix([f * Y[0] / Y[2] + cx - u, f * Y[1] / Y[2] + cy - v])
    r_depth = sp.Matrix([Y[2] - d])
    r_gp = sp.Matrix([sp.Matrix(n).dot(X) + h])
    rng = np.random.default_rng(5)
    for _ in range(5):
        pose = np.r_[synth.R_to_quat(synth.rot_vec(rng.normal(0, 0.4, 3))), rng.normal(0, 3, 3)]
        pose[:4] *= rng.uniform(0.9, 1.1)  # the polynomial R(q) is differentiated as it stands, not on the unit sphere
        camp = np.r_[synth.R_to_quat(synth.rot_vec(rng.normal(0, 0.4, 3))), rng.normal(0, 0.5, 3)]
        pt = np.array([rng.normal(0, 5), rng.normal(0, 3), rng.uniform(8, 40)])
        obs = dict(f=718.856, cx=607.19, cy=185.2, u=rng.uniform(0, 1200), v=rng.uniform(0, 370), d=rng.uniform(5, 40))
        nv, hv = rng.normal(0, 1, 3), rng.normal(0, 1)
        subs = dict(zip(q, pose)) | dict(zip(p, pt)) | dict(zip(c, camp)) | dict(zip(n, nv)) | {h: hv}
        subs |= {f: obs["f"], cx: obs["cx"], cy: obs["cy"], u: obs["u"], v: obs["v"], d: obs["d"]}
        for kind, expr, consts, blocks, params in (
            (0, r_repr, [obs["u"], obs["v"], obs["f"], obs["cx"], obs["cy"], *camp], (q, p), (pose, pt)),
            (1, r_depth, [obs["d"], *camp], (q, p), (pose, pt)),
            (3, r_gp, None, (q, n, (h,), p), (pose, nv, np.array([hv]), pt)),
        ):
            res, jacs = oracle.functor_jacobian(kind, consts, *params)
            assert np.allclose(res, np.array(expr.subs(subs), float).ravel(), rtol=1e-12, atol=1e-12)
            for blk, J in zip(blocks, jacs):
                Js = np.array(expr.jacobian(list(blk)).subs(subs), float)
                assert np.allclose(J, Js, rtol=1e-10, atol=1e-10), (kind, blk)


def test_sympy_local_jacobians_match_problem_evaluate(oracle):
    """Problem::Evaluate's Jacobians (robust_solving.cpp:44) are with respect to the manifold's tangent: sympy derivative of
    r(Plus(x, delta)) at delta = 0, Plus = quaternion left-multiplication by exp(delta) (Ceres QuaternionParameterization)."""
    sp, q, p, c, X, Y = _sym_setup()
    eps = sp.Symbol("eps", real=True)
    f, cx, cy = sp.symbols("f cx cy", real=True)
    r3 = sp.Matrix([f * Y[0] / Y[2] + cx, f * Y[1] / Y[2] + cy, Y[2]])  # the measurements are constants: they drop out
    w_, x_, y_, z_ = q[:4]
    cols = []
    for axis in range(6):
        if axis < 3:  # along a rotation axis delta = eps e_k: exp(delta) = (cos eps, sin eps e_k) exactly
            dq = [sp.cos(eps), 0, 0, 0]
            dq[1 + axis] = sp.sin(eps)
            qn = [dq[0] * w_ - dq[1] * x_ - dq[2] * y_ - dq[3] * z_, dq[0] * x_ + dq[1] * w_ + dq[2] * z_ - dq[3] * y_,
                  dq[0] * y_ - dq[1] * z_ + dq[2] * w_ + dq[3] * x_, dq[0] * z_ + dq[1] * y_ - dq[2] * x_ + dq[3] * w_]
            plus = dict(zip(q, qn + list(q[4:])))
        else:
            plus = {q[axis + 1]: q[axis + 1] + eps}
        cols.append(sp.diff(r3.subs(plus, simultaneous=True), eps).subs(eps, 0))
    Jpose = sp.Matrix.hstack(*cols)
    Jlm = r3.jacobian(list(p))
    fn_pose = sp.lambdify([q, p, c, f, cx, cy], Jpose, "numpy")
    fn_lm = sp.lambdify([q, p, c, f, cx, cy], Jlm, "numpy")
    w = synth.make_window(3, n_kf=3, n_lm=40, depth_prob=1.0)
    _, res, jp, jl, valid = oracle.evaluate(w, default_options(), apply_loss=False)
    assert valid.all()
    for i in range(w.n_obs):
        k, l, cc = w.obs_kf[i], w.obs_lm[i], w.obs_cam[i]
        args = (w.kf_pose[k], w.lm_pos[l], w.cam[cc, 3:10], w.cam[cc, 0], w.cam[cc, 1], w.cam[cc, 2])
        assert np.allclose(jp[i], np.array(fn_pose(*args), float), rtol=1e-9, atol=1e-9)
        assert np.allclose(jl[i], np.array(fn_lm(*args), float), rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------------------------------------------------- properties
window_shapes = st.tuples(st.integers(0, 10_000), st.integers(2, 5), st.integers(6, 40), st.sampled_from([0.0, 0.3, 1.0]), st.sampled_from([0.0, 0.3]))


@settings(max_examples=25, deadline=None, derandomize=True)
@given(window_shapes, st.integers(0, 2 ** 31 - 1))
def test_residuals_do_not_depend_on_the_origin_frame(oracle, shape, gseed):
    """Gauge: with X_k -> X_k G^-1 and p -> G p every keyframe-frame point X_k p is unchanged, so reprojection, depth and
    ground-height residuals are.  (The oracle's R(q) is the un-normalised polynomial: G and the poses are unit here.)"""
    seed, n_kf, n_lm, depth_prob, ground_frac = shape
    w = synth.make_window(seed, n_kf=n_kf, n_lm=n_lm, depth_prob=depth_prob, ground_frac=ground_frac)
    o = default_options()
    rng = np.random.default_rng(gseed)
    G_R, G_t = synth.rot_vec(rng.normal(0, 1.0, 3)), rng.normal(0, 20, 3)
    w2 = w.copy()
    for k in range(w.n_kf):
        R, t = synth.pose_to_Rt(w.kf_pose[k])
        w2.kf_pose[k] = synth.Rt_to_pose(R @ G_R.T, t - R @ G_R.T @ G_t)
    w2.lm_pos[:] = w.lm_pos @ G_R.T + G_t
    _, r1, _, _, v1 = oracle.evaluate(w, o, apply_loss=False)
    _, r2, _, _, v2 = oracle.evaluate(w2, o, apply_loss=False)
    assert (v1 == v2).all() and np.allclose(r1, r2, rtol=0, atol=1e-8 * max(1.0, np.abs(r1).max()))
    c1, _ = oracle.problem_cost(w, o)
    c2, _ = oracle.problem_cost(w2, o)
    assert abs(c1 - c2) <= 1e-9 * c1  # ground rows (nearest keyframe by |X_k p|) and regularisers (relative poses) included


@settings(max_examples=25, deadline=None, derandomize=True)
@given(window_shapes)
def test_schur_step_equals_the_dense_least_squares_step(oracle, shape):
    """SchurEliminator + dense Cholesky + back-substitution of the oracle against numpy's SVD-based lstsq on the stacked
    system [J; D] y = [r; 0] of the same first LM iteration."""
    seed, n_kf, n_lm, depth_prob, ground_frac = shape
    w = synth.make_window(seed, n_kf=n_kf, n_lm=n_lm, depth_prob=depth_prob, ground_frac=ground_frac)
    s = oracle.first_step(w, default_options())
    assume(s is not None)  # (a window whose solve takes no step: functor failure at x0 or convergence at iteration zero)
    J, r, D, y = s["J"], s["r"], s["D"], s["y"]
    A = np.vstack([J, np.diag(D)])
    y_dense = np.linalg.lstsq(A, np.r_[r, np.zeros(len(D))], rcond=None)[0]
    assert np.abs(y - y_dense).max() <= 1e-7 * max(1e-12, np.abs(y_dense).max())
