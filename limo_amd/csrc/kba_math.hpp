// kba_math.hpp — per-residual arithmetic of the keyframe-BA hot path, written for gfx950 lanes.
//
// Everything here is straight-line fp64 code with ANALYTIC Jacobians (no dual numbers on the hot path);
// one lane evaluates one observation.  Functions are __host__ __device__ so that the test-only serial
// emulation (tests/cpp/emu_pipeline.cpp) runs the same statements on the CPU.
//
// Reference being replaced (paths relative to the reference tree):
//   obs_residual_jacobian      ReprojectionErrorWithQuaternions + LandmarkDepthError evaluated through
//                              ceres::AutoDiffCostFunction<.,2|1,7,3>
//                              keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/internal/cost_functors_ceres.hpp:53-222
//                              created at keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:584-620
//   gp_residual_jacobian       GroundPlaneHeightRegularization, cost_functors_ceres.hpp:355-392, wiring
//                              bundle_adjuster_keyframes.cpp:517-562
//   loss_*                     ceres::CauchyLoss / HuberLoss / ScaledLoss + Corrector (Ceres 1.13) as used at
//                              bundle_adjuster_keyframes.cpp:553,589-591,616-618
//   pose_plus / unitvec_plus   ProductParameterization(QuaternionParameterization, Identity(3))
//                              (bundle_adjuster_keyframes.cpp:181-182) and FixScaleVectorPlus
//                              (internal/local_parameterizations.hpp:135-165)
// Pose convention: (qw,qx,qy,qz,tx,ty,tz), x' = R(q) x + t with Eigen's UN-NORMALISED polynomial R(q)
// (internal/definitions.hpp:75-83).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KBA_HD __host__ __device__ __forceinline__
#else
#define KBA_HD inline
#endif

namespace kba {

// ---------------------------------------------------------------------------------------- reciprocals on the hot path
// 1 / x and 1 / sqrt(x) for the solve's inner loops.  On gfx950 a correctly rounded fp64 division is ~12 instructions (63
// cycles of SIMD time at three waves, profiles/r04_micro_valu_f64_rate.txt), a square root ~18 (91 cycles): the IEEE
// expansions carry range scaling and fix-up steps for operands these call sites never see (x is a camera depth with
// |x| >= 0.01, a Cauchy denominator >= 1 or a positive pivot).  Here: the hardware seed (v_rcp_f64 / v_rsq_f64, ~23 bits)
// refined to the last one or two ulps - two Newton steps for the reciprocal, one third-order step for the inverse square
// root.  Not correctly rounded; every device path shares these, so the paths stay bit-identical among themselves, and the
// oracle bar (1e-4 on poses and cost, 1e-9 per Jacobian entry) is ten orders of magnitude away.  Host builds (the CPU-tier
// emulation) use the IEEE operations.
KBA_HD double rcp_nr(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / x;
#endif
}
KBA_HD double rsqrt_nr(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);          // 1 - x y^2
    return fma(y * e, fma(0.375, e, 0.5), y);        // y (1 + e/2 + 3 e^2/8): error^3
#else
    return 1.0 / sqrt(x);
#endif
}

// ---------------------------------------------------------------------------------------- rotation
// R(q) p and the 3x4 derivative d(R(q)p)/dq of the polynomial form (valid for non-unit q).
KBA_HD void quat_R(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.0 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.0 - (txx + tyy);
}

KBA_HD void mat3_vec(const double* R, const double* p, double* out) {
    out[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2];
    out[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2];
    out[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2];
}

// C = A B (3x3, row-major)
KBA_HD void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// M (3x3, row-major) = d(R(q)p)/dq (3x4) * dPlus/ddelta (4x3) of the left-multiplying quaternion update
// q (+) delta = [cos|d|, sin|d|/|d| d] (x) q.   For unit q this equals -2 [R p]_x.
KBA_HD void rot_tangent_jac(const double* q, const double* p, double* M) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double p0 = p[0], p1 = p[1], p2 = p[2];
    // columns of A = d(Rp)/d(w,x,y,z)
    const double Aw0 = 2.0 * (y * p2 - z * p1), Aw1 = 2.0 * (z * p0 - x * p2), Aw2 = 2.0 * (x * p1 - y * p0);
    const double Ax0 = 2.0 * (y * p1 + z * p2), Ax1 = 2.0 * (y * p0 - 2.0 * x * p1 - w * p2),
                 Ax2 = 2.0 * (z * p0 + w * p1 - 2.0 * x * p2);
    const double Ay0 = 2.0 * (-2.0 * y * p0 + x * p1 + w * p2), Ay1 = 2.0 * (x * p0 + z * p2),
                 Ay2 = 2.0 * (-w * p0 + z * p1 - 2.0 * y * p2);
    const double Az0 = 2.0 * (-2.0 * z * p0 - w * p1 + x * p2), Az1 = 2.0 * (w * p0 - 2.0 * z * p1 + y * p2),
                 Az2 = 2.0 * (x * p0 + y * p1);
    // P = [ -x -y -z ; w z -y ; -z w x ; y -x w ]  (QuaternionParameterization::ComputeJacobian)
    M[0] = -Aw0 * x + Ax0 * w - Ay0 * z + Az0 * y;
    M[1] = -Aw0 * y + Ax0 * z + Ay0 * w - Az0 * x;
    M[2] = -Aw0 * z - Ax0 * y + Ay0 * x + Az0 * w;
    M[3] = -Aw1 * x + Ax1 * w - Ay1 * z + Az1 * y;
    M[4] = -Aw1 * y + Ax1 * z + Ay1 * w - Az1 * x;
    M[5] = -Aw1 * z - Ax1 * y + Ay1 * x + Az1 * w;
    M[6] = -Aw2 * x + Ax2 * w - Ay2 * z + Az2 * y;
    M[7] = -Aw2 * y + Ax2 * z + Ay2 * w - Az2 * x;
    M[8] = -Aw2 * z - Ax2 * y + Ay2 * x + Az2 * w;
}

// The same M from R = R(q):  M(q, p) = -2 [Rh(q) p]_x  with the HOMOGENEOUS rotation polynomial Rh = R(q) + (|q|^2 - 1) I
// (quat_R is the form 1 - 2 (y^2 + z^2) ...: the two agree on unit quaternions) - a polynomial identity for every q, unit or not
// (tests/test_emu_vs_oracle.py::test_closed_form_rotation_jacobian_is_the_chain_rule_form holds the two statements against each
// other).  qq1 = |q|^2 - 1.  12 + 3 operations instead of ~100.
KBA_HD void rot_tangent_from_R(const double* R, double qq1, const double* p, double* M) {
    const double y0 = -2.0 * (R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + qq1 * p[0]);
    const double y1 = -2.0 * (R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + qq1 * p[1]);
    const double y2 = -2.0 * (R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + qq1 * p[2]);
    M[0] = 0.0, M[1] = -y2, M[2] = y1;
    M[3] = y2, M[4] = 0.0, M[5] = -y0;
    M[6] = -y1, M[7] = y0, M[8] = 0.0;
}
KBA_HD double quat_norm2_minus_1(const double* q) { return (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) - 1.0; }

// dR (3x3, row-major) = derivative of the polynomial R(q) along the tangent step delta of the left-multiplying quaternion
// update (dq = P(q) delta, P = QuaternionParameterization::ComputeJacobian): rot_tangent_jac(q, p, M) M delta == dR p
// for every p.  One matrix per keyframe instead of one M per observation wherever only the PRODUCT with a step is needed.
KBA_HD void quat_dR(const double* q, const double* delta, double* D) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double dw = -x * delta[0] - y * delta[1] - z * delta[2];
    const double dx = w * delta[0] + z * delta[1] - y * delta[2];
    const double dy = -z * delta[0] + w * delta[1] + x * delta[2];
    const double dz = y * delta[0] - x * delta[1] + w * delta[2];
    D[0] = -4.0 * (y * dy + z * dz);
    D[1] = 2.0 * (y * dx + x * dy - z * dw - w * dz);
    D[2] = 2.0 * (z * dx + x * dz + y * dw + w * dy);
    D[3] = 2.0 * (y * dx + x * dy + z * dw + w * dz);
    D[4] = -4.0 * (x * dx + z * dz);
    D[5] = 2.0 * (z * dy + y * dz - x * dw - w * dx);
    D[6] = 2.0 * (z * dx + x * dz - y * dw - w * dy);
    D[7] = 2.0 * (z * dy + y * dz + x * dw + w * dx);
    D[8] = -4.0 * (x * dx + y * dy);
}

// ---------------------------------------------------------------------------------------- projection into a view
// Camera-frame point z = H p + h0 of landmark p in a view with constants vl (H = Rc R(q) at [0..8], h0 = Rc t + tc at
// [9..11]; kba_items.hpp:view_consts_item).  ONE statement sequence with explicit fused multiply-adds: the linearisation and
// the kernels that rebuild the factored Jacobian from (au, sd) and the landmark (Schur fill, back-substitution) must get the
// same bits for the normalised coordinates xn = z0 / z2, yn = z1 / z2 whatever code surrounds the call.
template <class VP>
KBA_HD void view_point(VP vl, const double* p, double* z) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double t = vl[3 * i] * p[0];
        t = fma(vl[3 * i + 1], p[1], t);
        t = fma(vl[3 * i + 2], p[2], t);
        z[i] = t + vl[9 + i];
    }
}
// xn, yn and 1 / z of the point; a depth inside the functor's failure band (|z| < 0.01, cost_functors_ceres.hpp:78-83) is
// replaced by 1 - the caller's au and sd are zero there, the coordinates only have to stay finite.  Returns |z| >= 0.01.
template <class VP>
KBA_HD bool view_xy(VP vl, const double* p, double* xn, double* yn, double* iz_out = nullptr, double* z2_out = nullptr) {
    double z[3];
    view_point(vl, p, z);
    const bool z_ok = fabs(z[2]) >= 0.01;
    const double z2 = z_ok ? z[2] : 1.0;
    const double iz = rcp_nr(z2);
    *xn = z[0] * iz;
    *yn = z[1] * iz;
    if (iz_out) *iz_out = iz;
    if (z2_out) *z2_out = z2;
    return z_ok;
}

// ---------------------------------------------------------------------------------------- losses
// rho[0..2] of ScaledLoss(CauchyLoss(a), weight) at s
KBA_HD void loss_cauchy(double a, double weight, double s, double* rho) {
    const double b = a * a;
    const double c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = weight * (b * log(sum));
    rho[1] = weight * fmax(2.2250738585072014e-308, inv);
    rho[2] = weight * (-c * (inv * inv));
}
// rho' only (same arithmetic as loss_cauchy)
KBA_HD double loss_cauchy_d1(double a, double weight, double s) {
    const double b = a * a;
    const double c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    return weight * fmax(2.2250738585072014e-308, inv);
}
// ScaledLoss(HuberLoss(a), weight)
KBA_HD void loss_huber(double a, double weight, double s, double* rho) {
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        const double r1 = fmax(2.2250738585072014e-308, a / r);
        rho[0] = weight * (2.0 * a * r - b);
        rho[1] = weight * r1;
        rho[2] = weight * (-r1 / (2.0 * s));
    } else {
        rho[0] = weight * s;
        rho[1] = weight;
        rho[2] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------- observation
// One (keyframe, landmark, camera) measurement: reprojection rows 0,1 and (if d > 0) depth row 2.
//   pose   keyframe<-origin (7),  Rc/tc camera<-vehicle rotation (row-major 3x3) and translation
//   f,cx,cy intrinsics, (u,v,d) the measurement, lw landmark weight, a_rep/a_dep Cauchy scales.
// Outputs (all already multiplied by sqrt(rho') of the block's loss when apply_loss):
//   r[3]; Jp[18] = d r / d(rot tangent 3, translation 3) row-major 3x6; Jl[9] = d r / d landmark.
//   cost = 1/2 rho(|r_uv|^2) + 1/2 rho(r_d^2)    (un-robustified 1/2 |r|^2 when !apply_loss)
// Returns false where the reference functor fails (|z| < 0.01, cost_functors_ceres.hpp:78-83).
struct ObsOut {
    double r[3];
    double Jp[18];
    double Jl[9];
    double cost;
    double c[4];  // (au, xn, yn, sd): the translation columns of Jp are ft_build(c, Rc), see below
};

// Every row of an observation's Jacobian is  c_row^T Rc [ M(q,p) | I ]  (pose) /  c_row^T Rc R(q)  (landmark) with
//   c_u = au (1, 0, -xn),  c_v = au (0, 1, -yn),  c_d = sd (0, 0, 1),   au = sqrt(rho'_uv) f / z,  sd = sqrt(rho'_d).
// Ft = c^T Rc (3x3, = the translation columns of Jp) is rebuilt from the FOUR scalars and the view's Rc.
// (written as  au Rc_0 - (au xn) Rc_2: every multiply-add has ONE operand from Rc, which is wave-uniform in the landmark
// kernels - an instruction of gfx950 takes one scalar-register operand, the form au (Rc_0 - xn Rc_2) needs two and a copy)
template <class CP>
KBA_HD void ft_build(const double* c, CP Rc, double* Ft) {
    const double a1 = c[0] * c[1], a2 = c[0] * c[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Ft[0 + j] = c[0] * Rc[0 + j] - a1 * Rc[6 + j];
        Ft[3 + j] = c[0] * Rc[3 + j] - a2 * Rc[6 + j];
        Ft[6 + j] = c[3] * Rc[6 + j];
    }
}

KBA_HD bool obs_residual(const double* pose, const double* Rc, const double* tc, double f, double cx, double cy,
                         const double* lm, float u, float v, float d, double* r_uv, double* r_d, double* zc_out) {
    double R[9], y[3], zc[3];
    quat_R(pose, R);
    mat3_vec(R, lm, y);
    y[0] += pose[4];
    y[1] += pose[5];
    y[2] += pose[6];
    mat3_vec(Rc, y, zc);
    zc[0] += tc[0];
    zc[1] += tc[1];
    zc[2] += tc[2];
    if (zc_out) {
        zc_out[0] = zc[0];
        zc_out[1] = zc[1];
        zc_out[2] = zc[2];
    }
    *r_d = (d > 0.0f) ? zc[2] - static_cast<double>(d) : 0.0;
    if (!(fabs(zc[2]) >= 0.01)) return false;
    r_uv[0] = f * (zc[0] / zc[2]) + cx - static_cast<double>(u);
    r_uv[1] = f * (zc[1] / zc[2]) + cy - static_cast<double>(v);
    return true;
}

// want_cost = false skips the cost value (two logarithms): after an accepted step the cost at the new linearisation
// point is the candidate cost the step evaluation has already produced.
KBA_HD bool obs_residual_jacobian(const double* pose, const double* Rc, const double* tc, double f, double cx,
                                  double cy, const double* lm, float u, float v, float d, double lw, double a_rep,
                                  double a_dep, bool apply_loss, ObsOut* o, bool want_cost = true) {
    double R[9], Rp[3], y[3], zc[3];
    quat_R(pose, R);
    mat3_vec(R, lm, Rp);
    y[0] = Rp[0] + pose[4];
    y[1] = Rp[1] + pose[5];
    y[2] = Rp[2] + pose[6];
    mat3_vec(Rc, y, zc);
    zc[0] += tc[0];
    zc[1] += tc[1];
    zc[2] += tc[2];
    if (!(fabs(zc[2]) >= 0.01)) return false;
    const double iz = 1.0 / zc[2];
    const double xn = zc[0] * iz, yn = zc[1] * iz;
    const double ru = f * xn + cx - static_cast<double>(u);
    const double rv = f * yn + cy - static_cast<double>(v);
    const bool has_d = d > 0.0f;
    const double rd = has_d ? zc[2] - static_cast<double>(d) : 0.0;
    // d zc / d(rot tangent) = Rc * M ; d zc / dt = Rc ; d zc / d lm = Rc * R
    double M[9], G[9], H[9];
    rot_tangent_jac(pose, lm, M);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            G[i * 3 + j] = Rc[i * 3 + 0] * M[0 + j] + Rc[i * 3 + 1] * M[3 + j] + Rc[i * 3 + 2] * M[6 + j];
            H[i * 3 + j] = Rc[i * 3 + 0] * R[0 + j] + Rc[i * 3 + 1] * R[3 + j] + Rc[i * 3 + 2] * R[6 + j];
        }
    // projection rows: du = f/z (dx - xn dz), dv = f/z (dy - yn dz), dd = dz
    const double fz = f * iz;
    double su = 1.0, sd = 1.0, cost;
    const double s_uv = ru * ru + rv * rv;
    const double s_d = rd * rd;
    if (apply_loss) {
        double rho[3];
        if (want_cost) {
            loss_cauchy(a_rep, lw, s_uv, rho);
            su = sqrt(rho[1]);  // corrector: rho'' <= 0 for Cauchy  ->  plain sqrt(rho') scaling
            cost = 0.5 * rho[0];
            if (has_d) {
                loss_cauchy(a_dep, lw, s_d, rho);
                sd = sqrt(rho[1]);
                cost += 0.5 * rho[0];
            }
        } else {
            su = sqrt(loss_cauchy_d1(a_rep, lw, s_uv));
            if (has_d) sd = sqrt(loss_cauchy_d1(a_dep, lw, s_d));
            cost = 0.0;
        }
    } else {
        cost = 0.5 * s_uv + 0.5 * s_d;
    }
    if (!has_d) sd = 0.0;
    o->cost = cost;
    o->r[0] = su * ru;
    o->r[1] = su * rv;
    o->r[2] = sd * rd;
    const double au = su * fz, ad = sd;
    o->c[0] = au;
    o->c[1] = xn;
    o->c[2] = yn;
    o->c[3] = ad;
    for (int j = 0; j < 3; ++j) {
        o->Jp[0 * 6 + j] = au * (G[0 + j] - xn * G[6 + j]);
        o->Jp[1 * 6 + j] = au * (G[3 + j] - yn * G[6 + j]);
        o->Jp[2 * 6 + j] = ad * G[6 + j];
        o->Jp[0 * 6 + 3 + j] = au * (Rc[0 + j] - xn * Rc[6 + j]);
        o->Jp[1 * 6 + 3 + j] = au * (Rc[3 + j] - yn * Rc[6 + j]);
        o->Jp[2 * 6 + 3 + j] = ad * Rc[6 + j];
        o->Jl[0 * 3 + j] = au * (H[0 + j] - xn * H[6 + j]);
        o->Jl[1 * 3 + j] = au * (H[3 + j] - yn * H[6 + j]);
        o->Jl[2 * 3 + j] = ad * H[6 + j];
    }
    return true;
}

// Cost only (candidate point evaluation).  Returns false on functor failure.
KBA_HD bool obs_cost(const double* pose, const double* Rc, const double* tc, double f, double cx, double cy,
                     const double* lm, float u, float v, float d, double lw, double a_rep, double a_dep,
                     double* cost) {
    double r_uv[2], r_d;
    if (!obs_residual(pose, Rc, tc, f, cx, cy, lm, u, v, d, r_uv, &r_d, nullptr)) return false;
    double rho[3];
    loss_cauchy(a_rep, lw, r_uv[0] * r_uv[0] + r_uv[1] * r_uv[1], rho);
    double c = 0.5 * rho[0];
    if (d > 0.0f) {
        loss_cauchy(a_dep, lw, r_d * r_d, rho);
        c += 0.5 * rho[0];
    }
    *cost = c;
    return true;
}

// ---------------------------------------------------------------------------------------- ground plane
// r = n . (R(q) p + t) + h with loss ScaledLoss(Huber(0.1), w).   F[10] = d r / d(rot 3, t 3, n tangent 3, h),
// E[3] = d r / d p, all scaled by sqrt(rho') when apply_loss.
struct GpOut {
    double r;
    double F[10];
    double E[3];
    double cost;
};

KBA_HD void unitvec_plus_jac(const double* n, double* P) {
    // d/d delta of (n + delta)/|n + delta| at delta = 0 :  (I - n n^T/|n|^2)/|n|
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double inorm = 1.0 / sqrt(nn);
    const double i3 = inorm / nn;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P[i * 3 + j] = (i == j ? inorm : 0.0) - n[i] * n[j] * i3;
}

KBA_HD void gp_residual_jacobian(const double* pose, const double* n, double h, const double* lm, double w,
                                 bool apply_loss, bool want_jac, GpOut* o) {
    double R[9], y[3];
    quat_R(pose, R);
    mat3_vec(R, lm, y);
    y[0] += pose[4];
    y[1] += pose[5];
    y[2] += pose[6];
    const double r = n[0] * y[0] + n[1] * y[1] + n[2] * y[2] + h;
    double sc = 1.0;
    if (apply_loss) {
        double rho[3];
        loss_huber(0.1, w, r * r, rho);
        sc = sqrt(rho[1]);
        o->cost = 0.5 * rho[0];
    } else {
        o->cost = 0.5 * r * r;
    }
    o->r = sc * r;
    if (!want_jac) return;
    double M[9], P[9];
    rot_tangent_jac(pose, lm, M);
    unitvec_plus_jac(n, P);
    for (int j = 0; j < 3; ++j) {
        o->F[j] = sc * (n[0] * M[0 + j] + n[1] * M[3 + j] + n[2] * M[6 + j]);
        o->F[3 + j] = sc * n[j];
        o->F[6 + j] = sc * (y[0] * P[0 + j] + y[1] * P[3 + j] + y[2] * P[6 + j]);
        o->E[j] = sc * (n[0] * R[0 + j] + n[1] * R[3 + j] + n[2] * R[6 + j]);
    }
    o->F[9] = sc;
}

// ---------------------------------------------------------------------------------------- manifolds
KBA_HD void pose_plus(const double* x, const double* delta, double* out) {
    const double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (nd > 0.0) {
        const double sbd = sin(nd) / nd;
        const double z0 = cos(nd), z1 = sbd * delta[0], z2 = sbd * delta[1], z3 = sbd * delta[2];
        const double w0 = x[0], w1 = x[1], w2 = x[2], w3 = x[3];
        out[0] = z0 * w0 - z1 * w1 - z2 * w2 - z3 * w3;
        out[1] = z0 * w1 + z1 * w0 + z2 * w3 - z3 * w2;
        out[2] = z0 * w2 - z1 * w3 + z2 * w0 + z3 * w1;
        out[3] = z0 * w3 + z1 * w2 - z2 * w1 + z3 * w0;
    } else {
        out[0] = x[0];
        out[1] = x[1];
        out[2] = x[2];
        out[3] = x[3];
    }
    out[4] = x[4] + delta[3];
    out[5] = x[5] + delta[4];
    out[6] = x[6] + delta[5];
}

KBA_HD void unitvec_plus(const double* x, const double* delta, double* out) {
    const double a = x[0] + delta[0], b = x[1] + delta[1], c = x[2] + delta[2];
    const double factor = 1.0 / sqrt(a * a + b * b + c * c);
    out[0] = a * factor;
    out[1] = b * factor;
    out[2] = c * factor;
}

// ---------------------------------------------------------------------------------------- 3x3 SPD
// A (sym, 6 unique: a00 a01 a02 a11 a12 a22) = L L^T ; returns false if not positive definite.
// Linv (lower-triangular inverse, 6: l00 l10 l11 l20 l21 l22).
// Straight-line: the three pivots go through rsqrt_nr (what the factor's INVERSE needs is 1 / sqrt(pivot) - no square root
// followed by a division), a non-positive pivot is replaced by 1 and reported at the end.
KBA_HD bool chol3_inv(const double* A, double* Li) {
    const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    const bool ok0 = a00 > 0.0;
    const double i00 = rsqrt_nr(ok0 ? a00 : 1.0);
    const double l10 = a01 * i00, l20 = a02 * i00;
    const double d1 = a11 - l10 * l10;
    const bool ok1 = d1 > 0.0;
    const double i11 = rsqrt_nr(ok1 ? d1 : 1.0);
    const double l21 = (a12 - l20 * l10) * i11;
    const double d2 = a22 - l20 * l20 - l21 * l21;
    const bool ok2 = d2 > 0.0;
    const double i22 = rsqrt_nr(ok2 ? d2 : 1.0);
    Li[0] = i00;
    Li[1] = -l10 * i00 * i11;
    Li[2] = i11;
    Li[3] = -(l20 * i00 + l21 * Li[1]) * i22;
    Li[4] = -l21 * i11 * i22;
    Li[5] = i22;
    return ok0 && ok1 && ok2;
}

}  // namespace kba
