// kba_items.hpp — the work each lane / workgroup does in every kernel of the batched BA pipeline.
//
// Lane-level items are plain functions of (batch view, item index); cooperative (workgroup-per-window) items
// take (tid, nthreads) and a scratch pointer and use KBA_SYNC between phases, so the same statements run
//   * on gfx950 inside the __global__ wrappers of kba_kernels.hip (tid = threadIdx.x, scratch = LDS), and
//   * serially in tests/cpp/emu_pipeline.cpp (tid = 0, nthreads = 1) for CPU-side unit tests of the host logic.
// Reference rows: B1/B2 lin_obs / lin_lm_lane, B3 gp_lane, B4 cam_regs, B5 losses (kba_math.hpp),
// B6 manifolds, B7 trim_*, B8 lm_accum/lm_damp/schur_*/cam_assemble/cam_solve/backsub_lane.
#pragma once
#include "kba_layout.hpp"
#include "kba_lm.hpp"
#include "kba_math.hpp"

#ifndef KBA_SLAB_ENTRIES
#define KBA_SLAB_ENTRIES 4
#endif
#ifndef KBA_SYNC
#define KBA_SYNC() ((void)0)
#endif
// Phase stamps of the window-level items (debug builds of the HIP library with -DKBA_PROFILE_TICKS: the first and the
// last lane of window 0 record clock64() at every stamp, the kernel wrapper prints the differences).
#if defined(KBA_PROFILE_TICKS) && defined(__HIPCC__)
__device__ long long kba_ticks[2][24];
#endif
#if defined(KBA_PROFILE_TICKS) && defined(__HIP_DEVICE_COMPILE__)
#define KBA_TICK(n)                                                                   \
    do {                                                                              \
        if (w == 0 && (tid == 0 || tid == nt - 1)) kba_ticks[tid != 0][n] = clock64(); \
    } while (0)
#else
#define KBA_TICK(n) ((void)0)
#endif

namespace kba {

// ======================================================================================= observations
// Camera-side sums of one (landmark, view) pair in the RAW form the linearisation accumulates (kLinPartial = 28 entries, 26 used;
// lin_cam_half0 / lin_cam_half1 below): the camera assembly turns the per-view totals into U = Jp^T Jp and g = Jp^T r (cam_raw_entry).
struct LinLane {
    double e[kLinPartial];  // [0] cost | [1..25] raw sums | [26, 27] zero
    int fail;
};

// Per-view constants of the current poses (one item per view, before the observations are linearised):
//   vl[0..8] H = Rc R(q), vl[9..11] h0 = Rc t + tc (camera point = H p + h0), vl[12..20] Rc, vl[21..24] q, vl[25..27] f, cx, cy,
//   vl[28..36] R2 = -2 Rh(q) (row-major), Rh = R(q) + (|q|^2 - 1) I the homogeneous form of the rotation polynomial.  The
//   rotation-tangent Jacobian M(q, p) = d(R(q) p)/d(delta) of kba_math.hpp:rot_tangent_jac is the polynomial identity
//   M = -2 [Rh(q) p]_x  (for every q, unit or not), so with  y = R2 p  (9 multiply-adds)  M = [y]_x  and the camera-frame Jacobian  d(camera point)/d(rotation tangent) =
//   Rc M  has the entries  G[i][j] = Rc[i][j+1] y[j+2] - Rc[i][j+2] y[j+1]  (indices mod 3: 18 more) - 27 operations with
//   wave-uniform operands like the 27 of the per-view matrices C_k = Rc M(q, e_k) that rounds 5's first sessions used, but 9
//   constants per view instead of 27 (a pair's view constants are 37 doubles instead of 55: what the lane waits for in k_lin_lm).
// In k_lin_lm every lane of a wave is at the same view of the same window, so these are wave-uniform (scalar registers).
KBA_HD void view_consts_compute(const double* cam, const double* pose, double* vl);
KBA_HD void view_consts_item(const BatchView& bv, int view) {
    view_consts_compute(bv.view_cam + 16 * (int64_t)view, bv.pose + 7 * (int64_t)bv.view_kf[view], bv.view_lin + (int64_t)kViewLin * view);
}
// cam = the view's 16 camera doubles (f, cx, cy, -, Rc 9, tc 3), pose = its keyframe's 7; vl receives the 37 constants
KBA_HD void view_consts_compute(const double* cam, const double* pose, double* vl) {
    double R[9];
    quat_R(pose, R);
    mat3_mul(cam + 4, R, vl);
    for (int i = 0; i < 3; ++i)
        vl[9 + i] = cam[4 + 3 * i] * pose[4] + cam[4 + 3 * i + 1] * pose[5] + cam[4 + 3 * i + 2] * pose[6] + cam[13 + i];
    for (int i = 0; i < 9; ++i) vl[12 + i] = cam[4 + i];
    for (int i = 0; i < 4; ++i) vl[21 + i] = pose[i];
    vl[25] = cam[0];
    vl[26] = cam[1];
    vl[27] = cam[2];
    const double qq1 = quat_norm2_minus_1(pose);
    for (int i = 0; i < 9; ++i) vl[28 + i] = -2.0 * (R[i] + ((i & 3) == 0 ? qq1 : 0.0));  // (i = 0, 4, 8: the diagonal)
}

// Inputs of one observation as the linearisation consumes them (a GPU lane fetches them one observation ahead).
struct LinIn {
    double p[3];  // landmark
    double w;     // landmark weight
    double sw;    // sqrt(w): sqrt(rho') = sqrt(w) / sqrt(1 + s / a^2) for the scaled Cauchy loss
    float u, v, d;
    int live;     // landmark in the problem
};

KBA_HD void lin_fetch(const BatchView& bv, int64_t o, int gl, LinIn& in) {
    in.live = bv.lm_state[gl] != 0;
    in.p[0] = bv.lm[3 * (int64_t)gl];
    in.p[1] = bv.lm[3 * (int64_t)gl + 1];
    in.p[2] = bv.lm[3 * (int64_t)gl + 2];
    in.w = bv.lm_weight[gl];
    in.sw = sqrt(in.w);
    in.u = bv.obs_u[o];
    in.v = bv.obs_v[o];
    in.d = bv.obs_d[o];
}

// One observation at the CURRENT parameters: residual r (3, loss-corrected), the four scalars c4 = (au, xn, yn, sd) of
// the factored Jacobian (kba_math.hpp:ft_build) and the camera-side sums U = Jp^T Jp, g = Jp^T r with
// Jp = Ft [M | I], Ft = c^T Rc (ASSIGNED to out, not added).  The arithmetic of obs_residual_jacobian (kba_math.hpp), laid
// out for the instruction stream of a gfx950 lane (round 5: k_lin_lm 690 -> 547 us per round of 4096 slots,
// profiles/r04_ / r05_rocprof_kernel_stats_bench_one_group*.txt; the static instruction mix of the view loop is in
// profiles/r05_experiment_lin_lm_occupancy.txt):
//   * every multiply-add takes at most ONE operand from the view's constants vl (wave-uniform: scalar registers, of which an
//     instruction reads one);
//   * the pose Jacobian's rotation block in closed form from the view's -2 Rh(q) and Rc (view_consts_item): Rc M(q, p) = Rc [-2 Rh p]_x;
//   * 1 / z and sqrt(rho') through rcp_nr / rsqrt_nr: sqrt(w / (1 + s c)) = sqrt(w) rsqrt(1 + s c), one seed + one
//     refinement instead of a division followed by a square root;
//   * the cost value (two logarithms) only where the LM loop reads it (want_cost: the first linearisation of a solve).
// BRANCH-FREE on purpose: a landmark that is out of the problem (!in.live) or a failing functor (|z| < 0.01, returns
// false when in.live) contributes zeros through a final mask - on the GPU the passes of a lane are then one
// straight-line stream and the compiler can keep the loads of the next pass in flight across the stores of this one
// (with divergent branches around them it drained the memory queue every pass).
// CAM = false: the view's keyframe has no free pose block (WinDesc::n_view_fixed0): its camera-side sums would be masked out of
// the camera system anyway, so the pose Jacobian and U / g are not formed (a third of the arithmetic of a pair).  The cost, the
// residual and the planes are the same statements either way.
// VP: pointer to the view's constants - plain memory, or the constant address space (scalar loads) in k_lin_lm.
// ---- camera-side sums of a pair, RAW form (round 6).  The pose Jacobian of a pair is  J = C [G | Rc]  with
//   C = [[au, 0, -a1], [0, au, -a2], [0, 0, sd]]  (a1 = au xn, a2 = au yn: the rows u, v, d over the camera-frame coordinates),
//   G = Rc [y]_x,  y = -2 Rh(q) p  (lin_pose_jac),
// so  U = J^T J = [G | Rc]^T W [G | Rc]  with  W = C^T C  (four distinct values)  and  g = J^T r = [G | Rc]^T v,  v = C^T r.
// Rc is the VIEW's, so it leaves the sums over the landmarks: the lane accumulates
//   [1..4]   wa = au^2, wb = au a1, wc = au a2, wd = a1^2 + a2^2 + sd^2     W = [[wa, 0, -wb], [0, wa, -wc], [-wb, -wc, wd]]
//   [5..7]   v = (au r0, au r1, sd r2 - a1 r0 - a2 r1)
//   [8..16]  A = W G  (3 x 3, row-major)
//   [17..22] G^T A  (the rotation-rotation block, upper triangle 00 01 02 11 12 22)
//   [23..25] G^T v  (the rotation part of g)
// and the camera assembly forms  U_rot,trans = A^T Rc,  U_trans,trans = Rc^T W Rc,  g_trans = Rc^T v  from the per-view totals
// (cam_raw_entry).  86 multiply-adds per pair instead of the 138 of J (57) + J^T J, J^T r (81), 26 sums instead of 28, and the 3 x 6
// Jacobian itself is never formed (36 registers).  Two halves of 14 entries: what the second needs of the first stays in CamTmp.
struct CamTmp {
    double G[9], A01[6], v[3], wb, wc, wd;
};
template <class VP>
KBA_HD void lin_cam_half0(VP vl, const double* p, const double* c4, const double* r3, double cost, CamTmp& t, double* out) {
    const double p0 = p[0], p1 = p[1], p2 = p[2];
    const double au = c4[0], sd = c4[3];
    const double a1 = au * c4[1], a2 = au * c4[2];
    const double yv[3] = {vl[28] * p0 + vl[29] * p1 + vl[30] * p2, vl[31] * p0 + vl[32] * p1 + vl[33] * p2, vl[34] * p0 + vl[35] * p1 + vl[36] * p2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        constexpr int kNext[3] = {1, 2, 0};
        const int j1 = kNext[j], j2 = kNext[j1];
        t.G[0 + j] = vl[12 + j1] * yv[j2] - vl[12 + j2] * yv[j1];
        t.G[3 + j] = vl[15 + j1] * yv[j2] - vl[15 + j2] * yv[j1];
        t.G[6 + j] = vl[18 + j1] * yv[j2] - vl[18 + j2] * yv[j1];
    }
    const double wa = au * au;
    t.wb = au * a1;
    t.wc = au * a2;
    t.wd = a1 * a1 + a2 * a2 + sd * sd;
    t.v[0] = au * r3[0];
    t.v[1] = au * r3[1];
    t.v[2] = sd * r3[2] - a1 * r3[0] - a2 * r3[1];
    out[0] = cost;
    out[1] = wa;
    out[2] = t.wb;
    out[3] = t.wc;
    out[4] = t.wd;
    out[5] = t.v[0];
    out[6] = t.v[1];
    out[7] = t.v[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t.A01[j] = wa * t.G[j] - t.wb * t.G[6 + j];
        t.A01[3 + j] = wa * t.G[3 + j] - t.wc * t.G[6 + j];
        out[8 + j] = t.A01[j];
        out[11 + j] = t.A01[3 + j];
    }
}
KBA_HD void lin_cam_half1(const CamTmp& t, double* out) {
    double A2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        A2[j] = t.wd * t.G[6 + j] - t.wb * t.G[j] - t.wc * t.G[3 + j];
        out[j] = A2[j];
    }
    // G^T A, upper triangle: (i, j) = sum_k G[k][i] A[k][j]
    int q = 3;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) out[q++] = t.G[i] * t.A01[j] + t.G[3 + i] * t.A01[3 + j] + t.G[6 + i] * A2[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) out[9 + i] = t.G[i] * t.v[0] + t.G[3 + i] * t.v[1] + t.G[6 + i] * t.v[2];
    out[12] = 0.0;
    out[13] = 0.0;
}
// Entry q (0..26: U upper triangle row-major over the six pose slots [rot 3 | trans 3] 21, g 6) of a view's camera block from
// the view's raw totals R (kLinPartial entries) and its Rc (3 x 3 row-major).
KBA_HD double cam_raw_entry(int q, const double* R, const double* Rc) {
    if (q >= 21) {
        const int i = q - 21;
        if (i < 3) return R[23 + i];
        const int j = i - 3;
        return Rc[j] * R[5] + Rc[3 + j] * R[6] + Rc[6 + j] * R[7];
    }
    int a = 0, rem = q;
    while (rem >= 6 - a) {
        rem -= 6 - a;
        ++a;
    }
    const int b = a + rem;
    if (b < 3) {  // rotation x rotation: G^T A (upper triangle 00 01 02 11 12 22)
        return R[17 + (a == 0 ? b : a == 1 ? 2 + b : 5)];
    }
    const int j = b - 3;
    if (a < 3)  // rotation x translation: (A^T Rc)[a][j]
        return R[8 + a] * Rc[j] + R[11 + a] * Rc[3 + j] + R[14 + a] * Rc[6 + j];
    // translation x translation: (Rc^T W Rc)[i][j],  W = [[wa, 0, -wb], [0, wa, -wc], [-wb, -wc, wd]]
    const int i = a - 3;
    const double wr0 = R[1] * Rc[j] - R[2] * Rc[6 + j];                    // (W Rc)[0][j]
    const double wr1 = R[1] * Rc[3 + j] - R[3] * Rc[6 + j];                // (W Rc)[1][j]
    const double wr2 = R[4] * Rc[6 + j] - R[2] * Rc[j] - R[3] * Rc[3 + j];  // (W Rc)[2][j]
    return Rc[i] * wr0 + Rc[3 + i] * wr1 + Rc[6 + i] * wr2;
}

// Jp of an observation, row-major 3 x 6: [Ft M | Ft] = c^T [G | Rc],  G = Rc M(q, p) = Rc [y]_x,  y = -2 Rh(q) p, from the view's
// constants and the scalars (au, xn, yn, sd) of the factored form (shared by the solve's linearisation and the materialised pass)
template <class VP>
KBA_HD void lin_pose_jac(VP vl, const double* p, double au, double xn, double yn, double sd, double* J) {
    const double p0 = p[0], p1 = p[1], p2 = p[2];
    const double a1 = au * xn, a2 = au * yn;
    const double yv[3] = {vl[28] * p0 + vl[29] * p1 + vl[30] * p2, vl[31] * p0 + vl[32] * p1 + vl[33] * p2, vl[34] * p0 + vl[35] * p1 + vl[36] * p2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        constexpr int kNext[3] = {1, 2, 0};
        const int j1 = kNext[j], j2 = kNext[j1];
        const double g0 = vl[12 + j1] * yv[j2] - vl[12 + j2] * yv[j1];
        const double g1 = vl[15 + j1] * yv[j2] - vl[15 + j2] * yv[j1];
        const double g2 = vl[18 + j1] * yv[j2] - vl[18 + j2] * yv[j1];
        J[0 + j] = au * g0 - a1 * g2;
        J[6 + j] = au * g1 - a2 * g2;
        J[12 + j] = sd * g2;
        J[3 + j] = au * vl[12 + j] - a1 * vl[18 + j];
        J[9 + j] = au * vl[15 + j] - a2 * vl[18 + j];
        J[15 + j] = sd * vl[18 + j];
    }
}
// lin_obs_core: residual r3, the scalars c4 = (au, xn, yn, sd) of the factored rows and the cost of a pair; lin_obs adds the
// camera-side raw sums (CAM) or the cost alone.
template <class VP = const double*>
KBA_HD bool lin_obs_core(VP vl, const SolveConsts& c, const LinIn& in, bool want_cost, double* r3, double* c4, double& cost_out);
template <bool CAM = true, class VP = const double*>
KBA_HD bool lin_obs(VP vl, const SolveConsts& c, const LinIn& in, bool want_cost, double* r3, double* c4, LinLane& out) {
    const bool ok = lin_obs_core(vl, c, in, want_cost, r3, c4, out.e[0]);
    if (CAM) {
        CamTmp t;
        lin_cam_half0(vl, in.p, c4, r3, out.e[0], t, out.e);
        lin_cam_half1(t, out.e + 14);
    }
    return ok;
}
template <class VP>
KBA_HD bool lin_obs_core(VP vl, const SolveConsts& c, const LinIn& in, bool want_cost, double* r3, double* c4, double& cost_out) {
    double xn, yn, iz, z2;  // (z2 = 1 inside the failure band: keeps the arithmetic finite; masked below)
    const bool z_ok = view_xy(vl, in.p, &xn, &yn, &iz, &z2);
    const bool ok = in.live != 0 && z_ok;
    const double ru = vl[25] * xn + (vl[26] - static_cast<double>(in.u));
    const double rv = vl[25] * yn + (vl[27] - static_cast<double>(in.v));
    const bool has_d = in.d > 0.0f;
    const double rd = has_d ? z2 - static_cast<double>(in.d) : 0.0;
    const double s_uv = ru * ru + rv * rv, s_d = rd * rd;
    // ScaledLoss(CauchyLoss(a), w): rho' = w / (1 + s / a^2)   (a sum beyond 1e300 - a residual of 1e150 - is clamped: the
    // seed of the inverse square root stays a number)
    const double sum_uv = fmin(1.0 + s_uv * (1.0 / (c.a_rep * c.a_rep)), 1e300);
    const double sum_d = fmin(1.0 + s_d * (1.0 / (c.a_dep * c.a_dep)), 1e300);
    double su = in.sw * rsqrt_nr(sum_uv);
    double sd = in.sw * rsqrt_nr(sum_d);
    double cost = 0.0;
    if (want_cost) {  // (uniform over a workgroup)  1/2 rho(s) = 1/2 w a^2 log(1 + s / a^2)
        cost = 0.5 * (in.w * ((c.a_rep * c.a_rep) * log(sum_uv)));
        cost += has_d ? 0.5 * (in.w * ((c.a_dep * c.a_dep) * log(sum_d))) : 0.0;
    }
    su = ok ? su : 0.0;
    sd = ok && has_d ? sd : 0.0;
    cost_out = ok ? cost : 0.0;
    const double r0 = su * ru, r1 = su * rv, r2 = sd * rd;
    r3[0] = r0;
    r3[1] = r1;
    r3[2] = r2;
    const double au = su * (vl[25] * iz);
    c4[0] = au;  // (au and sd are what is stored per observation; xn, yn are rebuilt by the consumers: view_xy)
    c4[1] = xn;
    c4[2] = yn;
    c4[3] = sd;
    return z_ok || in.live == 0;
}

// ---- the MATERIALISED evaluation of one observation (Problem::Evaluate semantics; k_evaluate, SURVEY 8d's graded pass): residual
// r (3), J_pose (3 x 6 row-major, tangent: rotation 3 | translation 3), J_point (3 x 3) and the cost, all multiplied by sqrt(rho') of
// the block's loss when apply_loss - ReprojectionErrorWithQuaternions + LandmarkDepthError (cost_functors_ceres.hpp:53-222) through
// the statements of the solve's linearisation: view constants, rcp_nr / rsqrt_nr, the closed-form rotation block.  Without the loss:
// sqrt(rho') = 1, cost = 1/2 |r|^2.  Returns false where the functor fails (|z| < 0.01); the outputs are zero then.
struct EvalOut {
    double r[3], Jp[18], Jl[9], cost;
};
// Two stages, so that a kernel can finish with everything it LOADED (landmark, weight, measurement) before it issues its first store
// (gfx950 counts loads and stores in one counter: a load waited for behind a store waits for the store's write acknowledge too):
//   eval_obs_head: projection, residual, loss -> the scalars (au, xn, yn, sd) of the factored rows, r, cost
//   eval_obs_rows: J_pose, J_point from the scalars, the landmark and the view's constants
struct EvalHead {
    double au, xn, yn, sd, r[3], cost;
    bool ok;
};
template <class VP>
KBA_HD void eval_obs_head(VP vl, const SolveConsts& c, const double* p, double w, float u, float v, float d, bool apply_loss, EvalHead& h) {
    double xn, yn, iz, z2;
    const bool ok = view_xy(vl, p, &xn, &yn, &iz, &z2);
    const double ru = vl[25] * xn + (vl[26] - static_cast<double>(u));
    const double rv = vl[25] * yn + (vl[27] - static_cast<double>(v));
    const bool has_d = d > 0.0f;
    const double rd = has_d ? z2 - static_cast<double>(d) : 0.0;
    const double s_uv = ru * ru + rv * rv, s_d = rd * rd;
    double su = 1.0, sd = 1.0, cost;
    if (apply_loss) {  // (uniform over the launch)
        const double sum_uv = fmin(1.0 + s_uv * c.inv_a_rep2, 1e300);
        const double sum_d = fmin(1.0 + s_d * c.inv_a_dep2, 1e300);
        const double sw = sqrt(w);
        su = sw * rsqrt_nr(sum_uv);
        sd = sw * rsqrt_nr(sum_d);
        cost = 0.5 * (w * ((c.a_rep * c.a_rep) * log(sum_uv)));
        cost += has_d ? 0.5 * (w * ((c.a_dep * c.a_dep) * log(sum_d))) : 0.0;
    } else {
        cost = 0.5 * s_uv + 0.5 * s_d;
    }
    su = ok ? su : 0.0;
    sd = ok && has_d ? sd : 0.0;
    h.cost = ok ? cost : 0.0;
    h.r[0] = su * ru;
    h.r[1] = su * rv;
    h.r[2] = sd * rd;
    h.au = su * (vl[25] * iz);
    h.xn = xn;
    h.yn = yn;
    h.sd = sd;
    h.ok = ok;
}
template <class VP>
KBA_HD void eval_obs_rows(VP vl, const double* p, const EvalHead& h, double* Jp, double* Jl) {
    lin_pose_jac(vl, p, h.au, h.xn, h.yn, h.sd, Jp);
    const double c4[4] = {h.au, h.xn, h.yn, h.sd};
    ft_build(c4, vl, Jl);  // E = c^T H, H = Rc R(q)
}
template <class VP>
KBA_HD bool eval_obs(VP vl, const SolveConsts& c, const double* p, double w, float u, float v, float d, bool apply_loss, EvalOut& o) {
    EvalHead h;
    eval_obs_head(vl, c, p, w, u, v, d, apply_loss, h);
    for (int i = 0; i < 3; ++i) o.r[i] = h.r[i];
    o.cost = h.cost;
    eval_obs_rows(vl, p, h, o.Jp, o.Jl);
    return h.ok;
}

// ---- landmark-major linearisation (k_lin_lm): a lane holds ONE landmark and walks over the window's views.
// Per (landmark, view) pair: lin_obs as above (residual, factored Jacobian planes, camera-side sums), planes stored at
// the observation's index, and the landmark block V += E^T E, g += E^T r with E = c^T H - the statements of the
// former k_lm_accum, in the same view order, so V and g keep their bits; the 56 B / observation that kernel read back
// (and its launch) are gone.  The camera-side sums of a view leave the wave through one reduce-scatter per view
// (a slice per wave, no barrier inside the view loop; the four slices are added after the last view: BatchView::lv_part).
struct LmAcc {
    double V[6], g[3];
};
KBA_HD int lm_damp_store(const BatchView& bv, const SolveConsts& c, double radius, int gl, const double* s, const double* V, const double* g);
template <class VP>
KBA_HD void lin_lm_accum(VP vl, const double* r3, const double* c4, LmAcc& a) {
    double E[9];
    ft_build(c4, vl, E);  // E = c^T H, H = Rc R(q) of the view (view_consts_item)
    for (int row = 0; row < 3; ++row) {
        const double e0 = E[row * 3], e1 = E[row * 3 + 1], e2 = E[row * 3 + 2];
        a.V[0] += e0 * e0;
        a.V[1] += e0 * e1;
        a.V[2] += e0 * e2;
        a.V[3] += e1 * e1;
        a.V[4] += e1 * e2;
        a.V[5] += e2 * e2;
        a.g[0] += e0 * r3[row];
        a.g[1] += e1 * r3[row];
        a.g[2] += e2 * r3[row];
    }
}
// What the landmark's tail needs from memory besides its sums - fetched BEFORE the view loop (lin_lm_tail_fetch): behind the
// loop's plane stores these loads could not be moved up by the compiler (possible aliasing), and a dependent round trip per
// landmark at the end of a workgroup's life is a quarter of the kernel (profiles/r05_experiment_lin_lm_occupancy.txt: "no tail").
struct LmTailIn {
    int gg;             // ground-plane row of the landmark or -1
    int compute_scale;  // this linearisation defines the Jacobi scale
    double radius;      // trust-region radius of the coming step
    double sc[3];       // Jacobi scale of the landmark's columns (valid unless compute_scale)
};
KBA_HD void lin_lm_tail_fetch(const BatchView& bv, int w, int gl, LmTailIn& t) {
    t.gg = bv.lm_gp[gl];
    t.compute_scale = bv.st[w].compute_scale;
    t.radius = bv.st[w].radius;
    for (int i = 0; i < 3; ++i) t.sc[i] = t.compute_scale ? 1.0 : bv.lm_scale[i * bv.SL + gl];
}
// after the views: the landmark's ground-plane row, V / g / Jacobi scale to memory, damping.
// part: [0] max|g|, [1] |x|^2, [5] 1 = the damped block is not positive definite;  x = the landmark
KBA_HD void lin_lm_finish(const BatchView& bv, const SolveConsts& c, int gl, const double* x, const LmTailIn& t, LmAcc& a, double* part) {
    double* V = a.V;
    double* g = a.g;
    const int gg = t.gg;
    if (gg >= 0) {
        const double e0 = bv.gp_E[0 * bv.SG + gg], e1 = bv.gp_E[1 * bv.SG + gg], e2 = bv.gp_E[2 * bv.SG + gg];
        const double r = bv.gp_r[gg];
        V[0] += e0 * e0;
        V[1] += e0 * e1;
        V[2] += e0 * e2;
        V[3] += e1 * e1;
        V[4] += e1 * e2;
        V[5] += e2 * e2;
        g[0] += e0 * r;
        g[1] += e1 * r;
        g[2] += e2 * r;
    }
    for (int i = 0; i < 6; ++i) bv.lm_V[i * bv.SL + gl] = V[i];
    for (int i = 0; i < 3; ++i) bv.lm_g[i * bv.SL + gl] = g[i];
    double sc[3];
    if (t.compute_scale) {
        const double d[3] = {V[0], V[3], V[5]};
        for (int i = 0; i < 3; ++i) bv.lm_scale[i * bv.SL + gl] = sc[i] = c.jacobi_scaling ? 1.0 / (1.0 + sqrt(d[i])) : 1.0;
    } else {
        for (int i = 0; i < 3; ++i) sc[i] = t.sc[i];
    }
    part[0] = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
    part[1] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    // (V' + D^2) = L L^T for the step that follows this linearisation (the radius is final: kba_lm.hpp:lm_decide_step);
    // the stand-alone k_lm_damp only runs after rejected steps
    part[5] = lm_damp_store(bv, c, t.radius, gl, sc, V, g) ? 1.0 : 0.0;
}
// Plain form of one landmark for the CPU emulation (k_lin_lm runs the same statements software-pipelined and
// branch-free): cam[j] receives the camera-side sums of view j.  Returns 1 if a functor failed.
KBA_HD int lin_lm_lane(const BatchView& bv, const SolveConsts& c, int w, int gl, bool want_cost, LinLane* cam, double* part) {
    const WinDesc& wd = bv.win[w];
    part[0] = part[1] = part[5] = 0.0;
    for (int j = 0; j < wd.n_view; ++j) {
        cam[j].fail = 0;
        for (int i = 0; i < kLinPartial; ++i) cam[j].e[i] = 0.0;
    }
    const int state = bv.lm_state[gl];
    if (state == 0) return 0;
    LmAcc acc;
    for (int i = 0; i < 6; ++i) acc.V[i] = 0.0;
    for (int i = 0; i < 3; ++i) acc.g[i] = 0.0;
    int fail = 0;
    for (int j = 0; j < wd.n_view; ++j) {
        const int s = bv.lm_slot[(int64_t)j * bv.SL + gl];
        if (s < 0) continue;
        LinIn in;
        lin_fetch(bv, s, gl, in);
        const double* vl = bv.view_lin + (int64_t)kViewLin * (wd.view0 + j);
        double r3[3], c4[4];
        if (j < wd.n_view_fixed0) {  // keyframe without a free pose block: cost, residual and planes only
            if (!lin_obs<false>(vl, c, in, want_cost, r3, c4, cam[j])) fail = 1;
        } else {
            if (!lin_obs<true>(vl, c, in, want_cost, r3, c4, cam[j])) fail = 1;
        }
        bv.obs_c[s] = c4[0];  // (au, sd: the residual stays in the lane, xn / yn are rebuilt from the landmark: view_xy)
        bv.obs_c[bv.SO + s] = c4[3];
        lin_lm_accum(vl, r3, c4, acc);
    }
    if (state == 1) {
        LmTailIn t;
        lin_lm_tail_fetch(bv, w, gl, t);
        lin_lm_finish(bv, c, gl, bv.lm + 3 * (int64_t)gl, t, acc, part);
    }
    return fail;
}

// Un-robustified residual norms for trimming (robust_solving.cpp:24-44 with apply_loss = false).
KBA_HD void trim_residual_lane(const BatchView& bv, int b, int t, double* plane_rep, double* plane_dep) {
    if (t >= bv.blk_n[b]) return;
    const int view = bv.blk_view[b];
    const int64_t o = bv.blk_obs0[b] + t;
    const int gl = bv.obs_lm[o];
    double nr = -1.0, nd = -1.0;
    if (bv.lm_state[gl]) {
        const double* cam = bv.view_cam + 16 * (int64_t)view;
        double ruv[2], rd, zc[3];
        // the reprojection functor fails for |z| < 0.01 (its block then counts as an infinite residual); the
        // depth functor has no failure mode (cost_functors_ceres.hpp:193-212)
        if (obs_residual(bv.pose + 7 * (int64_t)bv.view_kf[view], cam + 4, cam + 13, cam[0], cam[1], cam[2],
                         bv.lm + 3 * (int64_t)gl, bv.obs_u[o], bv.obs_v[o], bv.obs_d[o], ruv, &rd, zc)) {
            nr = sqrt(ruv[0] * ruv[0] + ruv[1] * ruv[1]);
        } else {
            nr = INFINITY;
        }
        if (bv.obs_d[o] > 0.0f) nd = fabs(zc[2] - static_cast<double>(bv.obs_d[o]));
    }
    plane_rep[o] = nr;
    plane_dep[o] = nd;
}

// ======================================================================================= ground plane
KBA_HD void gp_lane(const BatchView& bv, int g, bool candidate, double* cost_out) {
    const int gl = bv.gp_lm[g];
    const int gk = bv.gp_kf[g];
    GpOut o;
    if (!bv.lm_state[gl]) {
        if (!candidate) {
            bv.gp_r[g] = 0.0;
            for (int i = 0; i < 10; ++i) bv.gp_F[i * bv.SG + g] = 0.0;
            for (int i = 0; i < 3; ++i) bv.gp_E[i * bv.SG + g] = 0.0;
        }
        cost_out[g] = 0.0;
        return;
    }
    if (candidate) {
        gp_residual_jacobian(bv.pose_c + 7 * (int64_t)gk, bv.pdir_c + 3 * (int64_t)gk, bv.pdist_c[gk],
                             bv.lm_c + 3 * (int64_t)gl, bv.gp_w[g], true, false, &o);
        cost_out[g] = o.cost;
        return;
    }
    gp_residual_jacobian(bv.pose + 7 * (int64_t)gk, bv.pdir + 3 * (int64_t)gk, bv.pdist[gk], bv.lm + 3 * (int64_t)gl,
                         bv.gp_w[g], true, true, &o);
    bv.gp_r[g] = o.r;
    for (int i = 0; i < 10; ++i) bv.gp_F[i * bv.SG + g] = o.F[i];
    for (int i = 0; i < 3; ++i) bv.gp_E[i * bv.SG + g] = o.E[i];
    cost_out[g] = o.cost;
}

// ======================================================================================= landmarks
// (V' + D^2) = L L^T with V' = S V S (Jacobi-scaled), D^2 = clamp(diag V')/radius.  Stores what the step needs of
// the landmark block:  Bt = L^-1 S  (lower triangular, 6: [l00 s0 | l10 s0, l11 s1 | l20 s0, l21 s1, l22 s2]) - every
// later use of L^-1 comes with the scale attached (Y' = .. S L^-T = .. Bt^T, delta = -S L^-T t = -Bt^T t) - and
// t = L^-1 S g = Bt g.  Returns 1 on Cholesky failure.
KBA_HD int lm_damp_store(const BatchView& bv, const SolveConsts& c, double radius, int gl, const double* s, const double* V, const double* g) {
    double A[6] = {s[0] * s[0] * V[0], s[0] * s[1] * V[1], s[0] * s[2] * V[2],
                   s[1] * s[1] * V[3], s[1] * s[2] * V[4], s[2] * s[2] * V[5]};
    const double ir = rcp_nr(radius);  // (one reciprocal instead of three divisions; radius > 0)
    A[0] += fmin(fmax(A[0], c.min_lm_diagonal), c.max_lm_diagonal) * ir;
    A[3] += fmin(fmax(A[3], c.min_lm_diagonal), c.max_lm_diagonal) * ir;
    A[5] += fmin(fmax(A[5], c.min_lm_diagonal), c.max_lm_diagonal) * ir;
    double Li[6];
    int fail = 0;
    if (!chol3_inv(A, Li)) {
        fail = 1;
        for (int i = 0; i < 6; ++i) Li[i] = 0.0;
    }
    const double Bt[6] = {Li[0] * s[0], Li[1] * s[0], Li[2] * s[1], Li[3] * s[0], Li[4] * s[1], Li[5] * s[2]};
    for (int i = 0; i < 6; ++i) bv.lm_Li[i * bv.SL + gl] = Bt[i];
    (void)g;  // t = L^-1 S g = Bt g is formed by its readers from Bt and g (lm_t_of): 24 B per landmark less to write and read
    return fail;
}
// t = L^-1 S g of a landmark from its stored Bt and g (the rhs column of the Schur tile, the back-substitution)
KBA_HD void lm_t_of(const double* Bt, const double* g, double* t) {
    t[0] = Bt[0] * g[0];
    t[1] = Bt[1] * g[0] + Bt[2] * g[1];
    t[2] = Bt[3] * g[0] + Bt[4] * g[1] + Bt[5] * g[2];
}
// Stand-alone pass: only after a REJECTED step (new radius, same linearisation) - after a linearisation the landmark pass
// (lin_lm_finish) has damped with the radius of the coming step already.
KBA_HD int lm_damp_lane(const BatchView& bv, const SolveConsts& c, int w, int gl) {
    if (bv.lm_state[gl] != 1) return 0;
    double s[3], V[6], g[3];
    for (int i = 0; i < 3; ++i) s[i] = bv.lm_scale[i * bv.SL + gl];
    for (int i = 0; i < 6; ++i) V[i] = bv.lm_V[i * bv.SL + gl];
    for (int i = 0; i < 3; ++i) g[i] = bv.lm_g[i * bv.SL + gl];
    return lm_damp_store(bv, c, bv.st[w].radius, gl, s, V, g);
}

// ======================================================================================= Schur tiles
// The Schur complement only involves the FREE camera slots of a window; they are numbered compactly (cslot: full
// local slot -> compact index or -1), pose slots first.  A Schur tile row (landmark i, coordinate c') is
//   Z[3i+c'][col(r)] = Y'_i[r][c'],   Y'_i = S_c F^T E S_l L^-T   (r over the free slots),   Z[3i+c'][nfq] = t_i[c']
// with col(r) = r + (r >= nfq): the rhs rides along as one more column, so  Z^T Z  delivers  sum Y'Y'^T  AND
// sum Y' t  in one symmetric rank-k update.  Landmarks without a ground-plane row only fill pose columns.
//
// schur_pair_block: the 3 x 10 block of Y' that landmark gl contributes to keyframe kl (local index): all of the
// keyframe's views of the landmark (Ft = c^T Rc, F = Ft [M | I], E = Ft R rebuilt from the factored planes) plus its ground-plane
// row when that row is attached to kl.  Y[a*3 + c'] for local slot a; masked slots are left zero.
//   lmk = Bt = L^-1 S of the landmark (6, lower, row-major; lm_damp_lane);  sc = Jacobi scale of the window's slots
//   (local index);  vkl[j] = local keyframe of view j.
// Pose part of one observation:  Y[a*3 + c'] += sc[a] (F^T E)[a][c] Bt[c'][c]   for the six pose slots, with
// F^T E = [M^T G ; G],  G = Ft^T Ft R  (F = Ft [M | I], E = Ft R).
// ASSIGN: Y receives the block (the lean kernel: one view per keyframe) instead of accumulating it.
template <bool ASSIGN = false>
KBA_HD void schur_pose_block(const double* Ft, const double* R, const double* M, const double* lmk, const double* sc6,
                             double* Y) {
    // A = Ft^T Ft (symmetric)
    const double a00 = Ft[0] * Ft[0] + Ft[3] * Ft[3] + Ft[6] * Ft[6];
    const double a01 = Ft[0] * Ft[1] + Ft[3] * Ft[4] + Ft[6] * Ft[7];
    const double a02 = Ft[0] * Ft[2] + Ft[3] * Ft[5] + Ft[6] * Ft[8];
    const double a11 = Ft[1] * Ft[1] + Ft[4] * Ft[4] + Ft[7] * Ft[7];
    const double a12 = Ft[1] * Ft[2] + Ft[4] * Ft[5] + Ft[7] * Ft[8];
    const double a22 = Ft[2] * Ft[2] + Ft[5] * Ft[5] + Ft[8] * Ft[8];
    double G[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[0 + j] = a00 * R[j] + a01 * R[3 + j] + a02 * R[6 + j];
        G[3 + j] = a01 * R[j] + a11 * R[3 + j] + a12 * R[6 + j];
        G[6 + j] = a02 * R[j] + a12 * R[3 + j] + a22 * R[6 + j];
    }
    // landmark-side factor first:  GB[i][c'] = sum_c G[i][c] Bt[c'][c]  (Bt lower triangular) - then the translation slots'
    // rows are the rows of GB and the rotation slots' rows are M^T GB: 18 + 27 multiply-adds instead of 27 + 36
    const double b00 = lmk[0], b01 = lmk[1], b02 = lmk[3];
    const double b11 = lmk[2], b12 = lmk[4];
    const double b22 = lmk[5];
    double GB[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        GB[i * 3 + 0] = G[i * 3] * b00;
        GB[i * 3 + 1] = G[i * 3] * b01 + G[i * 3 + 1] * b11;
        GB[i * 3 + 2] = G[i * 3] * b02 + G[i * 3 + 1] * b12 + G[i * 3 + 2] * b22;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const double yr = sc6[a] * (M[a] * GB[cc] + M[3 + a] * GB[3 + cc] + M[6 + a] * GB[6 + cc]);
            const double yt = sc6[3 + a] * GB[a * 3 + cc];
            Y[a * 3 + cc] = ASSIGN ? yr : Y[a * 3 + cc] + yr;
            Y[(3 + a) * 3 + cc] = ASSIGN ? yt : Y[(3 + a) * 3 + cc] + yt;
        }
    }
}

// Ground-plane row of landmark gl (row gg, attached to the keyframe whose free mask is cm / scale sc10).
KBA_HD void schur_gp_block(const BatchView& bv, int gg, const uint8_t* cm, const double* lmk, const double* sc10, double* Y) {
    double E[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) E[i] = bv.gp_E[i * bv.SG + gg];
    const double y0 = E[0] * lmk[0], y1 = E[0] * lmk[1] + E[1] * lmk[2], y2 = E[0] * lmk[3] + E[1] * lmk[4] + E[2] * lmk[5];
#pragma unroll
    for (int a = 0; a < kCamSlots; ++a) {
        if (!cm[a]) continue;
        const double f = bv.gp_F[a * bv.SG + gg] * sc10[a];
        Y[a * 3 + 0] += f * y0;
        Y[a * 3 + 1] += f * y1;
        Y[a * 3 + 2] += f * y2;
    }
}

KBA_HD void schur_pair_block(const BatchView& bv, const WinDesc& wd, int gl, int kl, const double* lmk, const double* sc,
                             const int* vkl, bool with_plane, double* Y) {
    for (int i = 0; i < 3 * kCamSlots; ++i) Y[i] = 0.0;
    const int row0 = kl * kCamSlots;
    const uint8_t* cm = bv.cmask + (int64_t)wd.cam0 + row0;
    if (cm[0]) {  // pose block free (all six slots are masked together)
        const double* pose = bv.pose + 7 * (int64_t)(wd.kf0 + kl);
        double R[9], M[9];
        bool have = false;
        for (int j = 0; j < wd.n_view; ++j) {
            if (vkl[j] != kl) continue;
            const int s = bv.lm_slot[(int64_t)j * bv.SL + gl];
            if (s < 0) continue;
            if (!have) {
                quat_R(pose, R);
                rot_tangent_from_R(R, quat_norm2_minus_1(pose), bv.lm + 3 * (int64_t)gl, M);
                have = true;
            }
            double Ft[9], c4[4];
            c4[0] = bv.obs_c[s];
            c4[3] = bv.obs_c[bv.SO + s];
            view_xy(bv.view_lin + (int64_t)kViewLin * (wd.view0 + j), bv.lm + 3 * (int64_t)gl, &c4[1], &c4[2]);
            ft_build(c4, bv.view_cam + 16 * (int64_t)(wd.view0 + j) + 4, Ft);
            schur_pose_block(Ft, R, M, lmk, sc + row0, Y);
        }
    }
    if (!with_plane) return;
    const int gg = bv.lm_gp[gl];
    if (gg < 0 || bv.gp_kf[gg] - wd.kf0 != kl) return;
    schur_gp_block(bv, gg, cm, lmk, sc + row0, Y);
}

KBA_HD void schur_load_lm(const BatchView& bv, int gl, double* lmk) {
#pragma unroll
    for (int i = 0; i < 6; ++i) lmk[i] = bv.lm_Li[i * bv.SL + gl];
}

// Partial slabs of a window: a wave takes `span` consecutive plain Schur blocks or `span_gp` consecutive ground-plane
// blocks and writes one slab; the slabs of the plain groups come first.
KBA_HD int schur_plain_slabs(const WinDesc& wd, int span) { return (wd.n_sblk_plain + span - 1) / span; }
KBA_HD int schur_slabs(const WinDesc& wd, int span, int span_gp) {
    return schur_plain_slabs(wd, span) + (wd.n_sblk - wd.n_sblk_plain + span_gp - 1) / span_gp;
}
KBA_HD int schur_slab_of(const WinDesc& wd, int sb, int span, int span_gp) {
    const int i = sb - wd.sblk0;
    return i < wd.n_sblk_plain ? i / span : schur_plain_slabs(wd, span) + (i - wd.n_sblk_plain) / span_gp;
}
// last block (inclusive) of the group that starts at block sb
KBA_HD int schur_group_last(const WinDesc& wd, int sb, int span, int span_gp) {
    const int i = sb - wd.sblk0;
    const bool plain = i < wd.n_sblk_plain;
    const int end = plain ? wd.n_sblk_plain : wd.n_sblk, sp = plain ? span : span_gp;
    return wd.sblk0 + (i + sp < end ? i + sp : end) - 1;
}

// The entries of [S | rhs] the camera solve reads: upper triangle of the nf free slots + the rhs, enumerated row by row
// (row ca: cb = ca .. nf, cb == nf is the rhs).  Entry i of that enumeration -> (ca, cb); its place in a full nfp x nfp slab.
KBA_HD int schur_need_count(int nf) { return nf * (nf + 1) / 2 + nf; }
KBA_HD int schur_need_pad(int nf) { return (schur_need_count(nf) + 31) / 32 * 32; }
KBA_HD void schur_need_decode(int i, int nf, int& ca, int& cb) {
    const int lda = nf + 1;  // row ca starts at ca * lda - ca (ca - 1) / 2
    ca = (int)(((2 * lda + 1) - sqrt((double)((2 * lda + 1) * (2 * lda + 1) - 8 * i))) * 0.5);
    while (ca > 0 && ca * lda - ca * (ca - 1) / 2 > i) --ca;
    while ((ca + 1) * lda - (ca + 1) * ca / 2 <= i) ++ca;
    cb = ca + (i - (ca * lda - ca * (ca - 1) / 2));
}
KBA_HD int schur_col(int i, int nfq);
KBA_HD int64_t schur_need_offset(int ca, int cb, int nf, int nfq, int nfp) {
    if (cb < nf) return (int64_t)schur_col(ca, nfq) * nfp + schur_col(cb, nfq);
    const int za = schur_col(ca, nfq);  // upper-triangle entry (za, nfq) or (nfq, za)
    return za < nfq ? (int64_t)za * nfp + nfq : (int64_t)nfq * nfp + za;
}
// Landmark-sharded solve: entry i (of the enumeration above) of the shard's contribution = sum over ALL partial slabs of the
// window in this shard's private S_part (slabs of other shards' workgroups are never written there and stay zero), in slab
// order.  bv.S_red of a producer view is the S part of the shard's own block (exchange_layout).
KBA_HD void slab_reduce_entry(const BatchView& bv, int w, int n_shards, int i) {
    const WinDesc& wd = bv.win[w];
    const int64_t slab = (int64_t)wd.nf_pad * wd.nf_pad;
    int ca, cb;
    schur_need_decode(i, wd.nf, ca, cb);
    const double* sp = bv.S_part + wd.spart_off + schur_need_offset(ca, cb, wd.nf, wd.nfq, wd.nf_pad);
    double a = 0.0;
    // (one slab per Schur block here; the plain blocks' slabs are zero - and since round 5 unwritten - outside the pose slots + rhs)
    const int q0 = (ca >= wd.nfq || (cb >= wd.nfq && cb < wd.nf)) ? wd.n_sblk_plain : 0;
    for (int q = q0; q < wd.n_sblk; ++q) a += sp[q * slab];
    bv.S_red[wd.sred_off / n_shards + i] = a;
}

KBA_HD int schur_col(int i, int nfq) {
    return i + (i >= nfq ? 1 : 0);
}

// ======================================================================================= back-substitution
// Product of factors >= 1 kept as mantissa x 2^exponent: sum_j log(f_j) = log(prod_j f_j) costs ONE logarithm per landmark
// and loss instead of one per observation (an fp64 logarithm is ~45 instructions on gfx950; the candidate cost of an
// observation is 1/2 w a^2 log(1 + s / a^2), twice).  The mantissa of n factors stays above 2^-n (n <= kMaxViews).
struct LogProd {
    double m;
    int e;
};
KBA_HD void logprod_mul(LogProd& a, double x) {
    int ex;
    a.m *= frexp(x, &ex);
    a.e += ex;
}
KBA_HD double logprod_log(const LogProd& a) { return log(a.m) + static_cast<double>(a.e) * 0.69314718055994530942; }

// y_l = (V'+D^2)^-1 (g' - W'^T y_c);  delta_l = -S_l y_l;  candidate = lm + delta_l;  then the cost of the landmark's
// observations AT THE CANDIDATE (Evaluator cost-only pass): the lane holds the candidate landmark, the candidate
// poses' per-view constants are in view_lin_c (k_cam_solve), so the measurements (12 B per observation through the slot
// table) are all that is read - the separate observation-major cost pass read 52 B per observation.
// part: [2] model-cost-change part, [3] |x - x_cand|^2, [4] |x_cand|^2, [6] candidate cost, [7] 1 = a functor failed
// w = window of the landmark (uniform over the workgroup: scalar loads of the window's and the views' constants)
KBA_HD void backsub_lane(const BatchView& bv, const SolveConsts& c, int w, int gl, double* part) {
    part[2] = part[3] = part[4] = part[6] = part[7] = 0.0;
    const int state = bv.lm_state[gl];
    const double* xp = bv.lm + 3 * (int64_t)gl;
    const double x[3] = {xp[0], xp[1], xp[2]};
    double xc[3] = {x[0], x[1], x[2]};
    if (state == 0) {  // out of the problem: the candidate is the point itself, no cost
        double* o = bv.lm_c + 3 * (int64_t)gl;
        o[0] = xc[0];
        o[1] = xc[1];
        o[2] = xc[2];
        return;
    }
    const WinDesc& wd = bv.win[w];
    // The view loops are branch-free and software-pipelined (slot two views ahead, data one view ahead): a pair the
    // landmark does not have reads observation 0 and contributes exact zeros - with `continue` around the loads the
    // memory queue drained at every view.
    const int32_t* slot = bv.lm_slot + gl;
    const int n_view = wd.n_view;
    if (state == 1) {
        double a[3] = {0, 0, 0};
        // (the leading views of keyframes without a free pose block have a zero camera step: their terms of `a` are exact
        // zeros - the loop starts behind them and their planes are never loaded)
        const int j0 = wd.n_view_fixed0;
        int s_cur = j0 < n_view ? slot[(int64_t)j0 * bv.SL] : -1;
        int s_nxt = j0 + 1 < n_view ? slot[(int64_t)(j0 + 1) * bv.SL] : -1;
        double aun, sdn;
        {
            const int64_t o = s_cur >= 0 ? s_cur : 0;
            aun = bv.obs_c[o];
            sdn = bv.obs_c[bv.SO + o];
        }
        for (int j = j0; j < n_view; ++j) {
            const bool have = s_cur >= 0;
            double c4[4];
            c4[0] = have ? aun : 0.0;
            c4[3] = have ? sdn : 0.0;
            s_cur = s_nxt;
            s_nxt = j + 2 < n_view ? slot[(int64_t)(j + 2) * bv.SL] : -1;
            {
                const int64_t o = s_cur >= 0 ? s_cur : 0;
                aun = bv.obs_c[o];
                sdn = bv.obs_c[bv.SO + o];
            }
            // E^T (F dc) of the pair, with F dc = c^T Rc (dR x + d_trans) = c^T (K x + k0) and E = c^T H: the view's
            // K = Rc dR, k0 = Rc d_trans (cam_solve) and H (view_consts_item) are wave-uniform, c^T is spanned by the four
            // scalars (kba_math.hpp:ft_build) - no 3 x 3 Ft / E per observation
            const double* vs = bv.view_lin_c + (int64_t)kViewLin * (wd.view0 + j) + 28;  // K (9) | k0 (3)
            const double* H = bv.view_lin + (int64_t)kViewLin * (wd.view0 + j);
            view_xy(H, x, &c4[1], &c4[2]);  // xn, yn of the pair as the linearisation had them
            const double w0 = vs[0] * x[0] + vs[1] * x[1] + vs[2] * x[2] + vs[9];
            const double w1 = vs[3] * x[0] + vs[4] * x[1] + vs[5] * x[2] + vs[10];
            const double w2 = vs[6] * x[0] + vs[7] * x[1] + vs[8] * x[2] + vs[11];
            const double v0 = c4[0] * (c4[0] * (w0 - c4[1] * w2));  // au q_u
            const double v1 = c4[0] * (c4[0] * (w1 - c4[2] * w2));  // au q_v
            const double v2 = c4[3] * (c4[3] * w2) - c4[1] * v0 - c4[2] * v1;
            a[0] += H[0] * v0 + H[3] * v1 + H[6] * v2;
            a[1] += H[1] * v0 + H[4] * v1 + H[7] * v2;
            a[2] += H[2] * v0 + H[5] * v1 + H[8] * v2;
        }
        const int gg = bv.lm_gp[gl];
        if (gg >= 0) {
            const int gk = bv.gp_kf[gg];
            const double* dc = bv.delta_c + (int64_t)gk * kCamSlots;
            double q = 0.0;
            for (int k = 0; k < kCamSlots; ++k) q += bv.gp_F[k * bv.SG + gg] * dc[k];
            for (int cc = 0; cc < 3; ++cc) a[cc] += bv.gp_E[cc * bv.SG + gg] * q;
        }
        double Bt[6], t[3], V[6], g[3];
        for (int i = 0; i < 6; ++i) Bt[i] = bv.lm_Li[i * bv.SL + gl];  // L^-1 S (lm_damp_lane)
        for (int i = 0; i < 6; ++i) V[i] = bv.lm_V[i * bv.SL + gl];
        for (int i = 0; i < 3; ++i) g[i] = bv.lm_g[i * bv.SL + gl];
        lm_t_of(Bt, g, t);
        // W'^T y_c = -S_l a  (a built from the UNSCALED camera step delta_c = -S_c y_c):  t' = t + L^-1 S a
        const double t0 = t[0] + Bt[0] * a[0];
        const double t1 = t[1] + Bt[1] * a[0] + Bt[2] * a[1];
        const double t2 = t[2] + Bt[3] * a[0] + Bt[4] * a[1] + Bt[5] * a[2];
        // delta_l = -S L^-T t'
        const double d0 = -(Bt[0] * t0 + Bt[1] * t1 + Bt[3] * t2);
        const double d1 = -(Bt[2] * t1 + Bt[4] * t2);
        const double d2 = -(Bt[5] * t2);
        xc[0] = x[0] + d0;
        xc[1] = x[1] + d1;
        xc[2] = x[2] + d2;
        const double Vd0 = V[0] * d0 + V[1] * d1 + V[2] * d2;
        const double Vd1 = V[1] * d0 + V[3] * d1 + V[4] * d2;
        const double Vd2 = V[2] * d0 + V[4] * d1 + V[5] * d2;
        part[2] = -(g[0] * d0 + g[1] * d1 + g[2] * d2) - (a[0] * d0 + a[1] * d1 + a[2] * d2) -
                  0.5 * (d0 * Vd0 + d1 * Vd1 + d2 * Vd2);
        const double e0 = x[0] - xc[0], e1 = x[1] - xc[1], e2 = x[2] - xc[2];
        part[3] = e0 * e0 + e1 * e1 + e2 * e2;
        part[4] = xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2];
    }
    // ---- cost of this landmark's observations at (candidate poses, candidate point); state 2 = constant landmark of a
    //      motion-only problem: its point is the candidate.  1/2 rho = 1/2 w a^2 log(1 + s / a^2) per block: the factors
    //      1 + s / a^2 of the landmark's blocks are multiplied up per loss (LogProd), two logarithms per LANDMARK.
    const double lw = bv.lm_weight[gl];
    const double c_rep = 1.0 / (c.a_rep * c.a_rep), c_dep = 1.0 / (c.a_dep * c.a_dep);
    LogProd lp_rep = {1.0, 0}, lp_dep = {1.0, 0};
    int fail = 0;
    {
        int s_cur = slot[0];
        int s_nxt = n_view > 1 ? slot[bv.SL] : -1;
        float un, vn, dn;
        {
            const int64_t o = s_cur >= 0 ? s_cur : 0;
            un = bv.obs_u[o];
            vn = bv.obs_v[o];
            dn = bv.obs_d[o];
        }
        for (int j = 0; j < n_view; ++j) {
            const bool have = s_cur >= 0;
            const float u = un, v = vn, d = dn;
            s_cur = s_nxt;
            s_nxt = j + 2 < n_view ? slot[(int64_t)(j + 2) * bv.SL] : -1;
            {
                const int64_t o = s_cur >= 0 ? s_cur : 0;
                un = bv.obs_u[o];
                vn = bv.obs_v[o];
                dn = bv.obs_d[o];
            }
            const double* vc = bv.view_lin_c + (int64_t)kViewLin * (wd.view0 + j);
            const double z0 = vc[0] * xc[0] + vc[1] * xc[1] + vc[2] * xc[2] + vc[9];
            const double z1 = vc[3] * xc[0] + vc[4] * xc[1] + vc[5] * xc[2] + vc[10];
            const double z2r = vc[6] * xc[0] + vc[7] * xc[1] + vc[8] * xc[2] + vc[11];
            const bool z_ok = fabs(z2r) >= 0.01;
            if (have && !z_ok) fail = 1;
            const double z2 = z_ok ? z2r : 1.0;  // keeps the arithmetic finite; masked below
            const double iz = rcp_nr(z2);
            const double ru = vc[25] * (z0 * iz) + (vc[26] - static_cast<double>(u));
            const double rv = vc[25] * (z1 * iz) + (vc[27] - static_cast<double>(v));
            const double rd = z2 - static_cast<double>(d);
            const bool on = have && z_ok;
            const double f_rep = fmin(1.0 + (ru * ru + rv * rv) * c_rep, 1e300);
            const double f_dep = fmin(1.0 + (rd * rd) * c_dep, 1e300);
            logprod_mul(lp_rep, on ? f_rep : 1.0);
            logprod_mul(lp_dep, on && d > 0.0f ? f_dep : 1.0);
        }
    }
    part[6] = 0.5 * (lw * ((c.a_rep * c.a_rep) * logprod_log(lp_rep))) + 0.5 * (lw * ((c.a_dep * c.a_dep) * logprod_log(lp_dep)));
    part[7] = fail ? 1.0 : 0.0;
    double* o = bv.lm_c + 3 * (int64_t)gl;
    o[0] = xc[0];
    o[1] = xc[1];
    o[2] = xc[2];
}

// ======================================================================================= camera-only rows
// Regulariser rows of a window (bundle_adjuster_keyframes.cpp:769-818, :890-904, :835-853; functors
// cost_functors_ceres.hpp:224-250, 300-353, 394-438, 507-555), analytic, as sparse rows over the window's camera
// slots: up to 16 (column, value) pairs per row.  Row residuals/Jacobians are multiplied by sqrt(weight)
// (ScaledLoss(TrivialLoss, w) => rho = w s, corrector = sqrt(w)).
struct RegRow {
    double r;
    int n;
    int col[16];
    double val[16];
};

KBA_HD void quat_apply_jac(const double* q, const double* p, double* A) {  // A[3][4] = d(R(q)p)/d(w,x,y,z)
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double p0 = p[0], p1 = p[1], p2 = p[2];
    A[0] = 2.0 * (y * p2 - z * p1);
    A[4] = 2.0 * (z * p0 - x * p2);
    A[8] = 2.0 * (x * p1 - y * p0);
    A[1] = 2.0 * (y * p1 + z * p2);
    A[5] = 2.0 * (y * p0 - 2.0 * x * p1 - w * p2);
    A[9] = 2.0 * (z * p0 + w * p1 - 2.0 * x * p2);
    A[2] = 2.0 * (-2.0 * y * p0 + x * p1 + w * p2);
    A[6] = 2.0 * (x * p0 + z * p2);
    A[10] = 2.0 * (-w * p0 + z * p1 - 2.0 * y * p2);
    A[3] = 2.0 * (-2.0 * z * p0 - w * p1 + x * p2);
    A[7] = 2.0 * (w * p0 - 2.0 * z * p1 + y * p2);
    A[11] = 2.0 * (x * p0 + y * p1);
}

// N (3x3) = d(R(q)^T v)/d(rot tangent):  R(q)^T = R(conj q) for the polynomial form.
KBA_HD void rotT_tangent_jac(const double* q, const double* v, double* N) {
    const double qc[4] = {q[0], -q[1], -q[2], -q[3]};
    double A[12];
    quat_apply_jac(qc, v, A);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    for (int i = 0; i < 3; ++i) {
        const double aw = A[i * 4 + 0], ax = -A[i * 4 + 1], ay = -A[i * 4 + 2], az = -A[i * 4 + 3];
        N[i * 3 + 0] = -aw * x + ax * w - ay * z + az * y;
        N[i * 3 + 1] = -aw * y + ax * z + ay * w - az * x;
        N[i * 3 + 2] = -aw * z - ax * y + ay * x + az * w;
    }
}

// d = t_a - R_a R_b^T t_b (translation of T_a * T_b^-1) and its derivatives wrt both tangents.
// Ja[3][6], Jb[3][6] (rot 3, trans 3)
KBA_HD void rel_translation(const double* pa, const double* pb, double* d, double* Ja, double* Jb, bool want_jac) {
    double Ra[9], Rb[9], u[3], Ru[3];
    quat_R(pa, Ra);
    quat_R(pb, Rb);
    // u = R_b^T t_b
    for (int i = 0; i < 3; ++i) u[i] = Rb[0 + i] * pb[4] + Rb[3 + i] * pb[5] + Rb[6 + i] * pb[6];
    mat3_vec(Ra, u, Ru);
    for (int i = 0; i < 3; ++i) d[i] = pa[4 + i] - Ru[i];
    if (!want_jac) return;
    double M[9], N[9];
    rot_tangent_jac(pa, u, M);
    rotT_tangent_jac(pb, pb + 4, N);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Ja[i * 6 + j] = -M[i * 3 + j];
            Ja[i * 6 + 3 + j] = (i == j) ? 1.0 : 0.0;
            Jb[i * 6 + j] = -(Ra[i * 3 + 0] * N[0 + j] + Ra[i * 3 + 1] * N[3 + j] + Ra[i * 3 + 2] * N[6 + j]);
            // -R_a R_b^T
            Jb[i * 6 + 3 + j] = -(Ra[i * 3 + 0] * Rb[j * 3 + 0] + Ra[i * 3 + 1] * Rb[j * 3 + 1] + Ra[i * 3 + 2] * Rb[j * 3 + 2]);
        }
}

// Number of regulariser rows of a window and evaluation of row `idx`.
// Row order: [scale reg] then per consecutive pair k0: normal diff (3), dist diff (1), motion (1); then per
// keyframe global normal (3); pose-only: speed prior (3).
// Entry (row, k) of a row-major 3x3 held in registers, `row` known only at run time: selects instead of an indexed
// read, which would push the whole matrix into scratch memory.
KBA_HD double sel_row3(const double* P, int row, int k) { return row == 0 ? P[k] : (row == 1 ? P[3 + k] : P[6 + k]); }

KBA_HD int reg_row_count(const WinDesc& wd) {
    int n = 0;
    if (wd.has_scale_reg) n += 1;
    if (wd.has_gp_reg) n += (wd.n_kf - 1) * 5 + wd.n_kf * 3;
    if (wd.pose_only && wd.speed_w > 0.0) n += 3;
    return n;
}

// Rows (in the numbering of reg_row_eval below) whose residual involves keyframes ka and kb (kb == ka or ka + 1), in
// ascending order: f(first row, number of rows, lowest keyframe of those rows) per run of rows.
template <class F>
KBA_HD void reg_rows_of_block(const WinDesc& wd, int ka, int kb, F&& f) {
    int i0 = 0;
    if (wd.has_scale_reg) {
        if (kb <= 1) f(0, 1, 0);  // PoseRegularization(pose 1, pose 0)
        i0 = 1;
    }
    if (wd.has_gp_reg) {
        const int npair = wd.n_kf - 1;
        if (ka == kb && ka >= 1) f(i0 + (ka - 1) * 5, 5, ka - 1);  // pair (ka - 1, ka)
        if (ka < npair) f(i0 + ka * 5, 5, ka);                      // pair (ka, ka + 1)
        if (ka == kb) f(i0 + npair * 5 + ka * 3, 3, ka);            // rows of keyframe ka alone
        i0 += npair * 5 + wd.n_kf * 3;
    }
    if (wd.pose_only && wd.speed_w > 0.0 && kb == 0) f(i0, 3, 0);
}

// Evaluates row idx at (pose, pdir, pdist) arrays indexed by GLOBAL keyframe.  Columns are LOCAL camera slots.
// all_const receives 1 if every parameter block of the row's residual block is constant (fixed cost).
KBA_HD void reg_row_eval(const WinDesc& wd, const uint8_t* cmask, const double* pose, const double* pdir,
                         const double* pdist, int idx, bool want_jac, RegRow& row, int& all_const) {
    row.n = 0;
    row.r = 0.0;
    all_const = 0;
    auto blk_free = [&](int kf_local, int slot) { return cmask[(int64_t)(wd.kf0 + kf_local) * kCamSlots + slot] != 0; };
    auto push = [&](int kf_local, int slot, double v) {
        row.col[row.n] = kf_local * kCamSlots + slot;
        row.val[row.n] = v;
        row.n++;
    };
    int i = idx;
    if (wd.has_scale_reg) {
        if (i == 0) {  // PoseRegularization(pose[1], pose[0]), weight scale_w
            const double sw = sqrt(wd.scale_w);
            double d[3], Ja[18], Jb[18];
            rel_translation(pose + 7 * (int64_t)(wd.kf0 + 1), pose + 7 * (int64_t)wd.kf0, d, Ja, Jb, want_jac);
            const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            row.r = sw * (nrm - wd.scale_s0);
            all_const = !blk_free(1, 0) && !blk_free(0, 0);
            if (want_jac) {
                const double e[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};
                for (int k = 0; k < 6; ++k) push(1, k, sw * (e[0] * Ja[k] + e[1] * Ja[6 + k] + e[2] * Ja[12 + k]));
                for (int k = 0; k < 6; ++k) push(0, k, sw * (e[0] * Jb[k] + e[1] * Jb[6 + k] + e[2] * Jb[12 + k]));
            }
            return;
        }
        i -= 1;
    }
    if (wd.has_gp_reg) {
        const int npair = wd.n_kf - 1;
        if (i < npair * 5) {
            const int k0 = i / 5, sub = i % 5, k1 = k0 + 1;
            const double* n0 = pdir + 3 * (int64_t)(wd.kf0 + k0);
            const double* n1 = pdir + 3 * (int64_t)(wd.kf0 + k1);
            if (sub < 3) {  // VectorDifferenceRegularization(n1, n0), weight 30
                const double sw = sqrt(30.0);
                row.r = sw * (n1[sub] - n0[sub]);
                all_const = !blk_free(k1, 6) && !blk_free(k0, 6);
                if (want_jac) {
                    double P1[9], P0[9];
                    unitvec_plus_jac(n1, P1);
                    unitvec_plus_jac(n0, P0);
                    for (int k = 0; k < 3; ++k) push(k1, 6 + k, sw * sel_row3(P1, sub, k));
                    for (int k = 0; k < 3; ++k) push(k0, 6 + k, -sw * sel_row3(P0, sub, k));
                }
            } else if (sub == 3) {  // GroundPlaneDistanceRegularization(h1, h0), weight 10
                const double sw = sqrt(10.0);
                row.r = sw * (pdist[wd.kf0 + k1] - pdist[wd.kf0 + k0]);
                all_const = !blk_free(k1, 9) && !blk_free(k0, 9);
                if (want_jac) {
                    push(k1, 9, sw);
                    push(k0, 9, -sw);
                }
            } else {  // GroundPlaneMotionRegularization(pose_k0, pose_k1, n_k0), weight 20
                const double sw = sqrt(20.0);
                double d[3], Ja[18], Jb[18];
                rel_translation(pose + 7 * (int64_t)(wd.kf0 + k0), pose + 7 * (int64_t)(wd.kf0 + k1), d, Ja, Jb, want_jac);
                const double z = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                double dn[3] = {d[0], d[1], d[2]};
                double inv = 1.0;
                if (z > 0.0) {
                    inv = 1.0 / sqrt(z);
                    for (int k = 0; k < 3; ++k) dn[k] *= inv;
                }
                row.r = sw * (n0[0] * dn[0] + n0[1] * dn[1] + n0[2] * dn[2]);
                all_const = !blk_free(k0, 0) && !blk_free(k1, 0) && !blk_free(k0, 6);
                if (want_jac) {
                    // d r / d d = n0^T (I - dn dn^T) / |d|   (identity when |d| == 0: normalize() is skipped)
                    double e[3];
                    if (z > 0.0) {
                        const double ndn = n0[0] * dn[0] + n0[1] * dn[1] + n0[2] * dn[2];
                        for (int k = 0; k < 3; ++k) e[k] = (n0[k] - ndn * dn[k]) * inv;
                    } else {
                        for (int k = 0; k < 3; ++k) e[k] = n0[k];
                    }
                    double P0[9];
                    unitvec_plus_jac(n0, P0);
                    for (int k = 0; k < 6; ++k) push(k0, k, sw * (e[0] * Ja[k] + e[1] * Ja[6 + k] + e[2] * Ja[12 + k]));
                    for (int k = 0; k < 6; ++k) push(k1, k, sw * (e[0] * Jb[k] + e[1] * Jb[6 + k] + e[2] * Jb[12 + k]));
                    for (int k = 0; k < 3; ++k)
                        push(k0, 6 + k, sw * (dn[0] * P0[0 + k] + dn[1] * P0[3 + k] + dn[2] * P0[6 + k]));
                }
            }
            return;
        }
        i -= npair * 5;
        if (i < wd.n_kf * 3) {  // VectorDifferenceRegularization2((0,0,1), n_k), weight 10
            const int k = i / 3, sub = i % 3;
            const double sw = sqrt(10.0);
            const double* n = pdir + 3 * (int64_t)(wd.kf0 + k);
            const double tgt[3] = {0.0, 0.0, 1.0};
            row.r = sw * (tgt[sub] - n[sub]);
            all_const = !blk_free(k, 6);
            if (want_jac) {
                double P[9];
                unitvec_plus_jac(n, P);
                for (int kk = 0; kk < 3; ++kk) push(k, 6 + kk, -sw * sel_row3(P, sub, kk));
            }
            return;
        }
        i -= wd.n_kf * 3;
    }
    if (wd.pose_only && wd.speed_w > 0.0 && i < 3) {
        // SpeedRegularizationVector2: r = (R_new (-R_b^T t_b) + t_new)/dt - vel_prev
        const double sw = sqrt(wd.speed_w);
        const double* p = pose + 7 * (int64_t)wd.kf0;
        double u[3], R[9], Ru[3];
        for (int k = 0; k < 3; ++k)
            u[k] = -(wd.speed_Rb[0 + k] * wd.speed_tb[0] + wd.speed_Rb[3 + k] * wd.speed_tb[1] + wd.speed_Rb[6 + k] * wd.speed_tb[2]);
        quat_R(p, R);
        mat3_vec(R, u, Ru);
        const double Rui = i == 0 ? Ru[0] : (i == 1 ? Ru[1] : Ru[2]);
        row.r = sw * ((Rui + p[4 + i]) / wd.speed_dt - wd.speed_vel[i]);
        all_const = !blk_free(0, 0);
        if (want_jac) {
            double M[9];
            rot_tangent_jac(p, u, M);
            for (int k = 0; k < 3; ++k) push(0, k, sw * sel_row3(M, i, k) / wd.speed_dt);
            for (int k = 0; k < 3; ++k) push(0, 3 + k, sw * ((i == k) ? 1.0 : 0.0) / wd.speed_dt);
        }
        return;
    }
}

// ======================================================================================= camera system
// |x - Plus(x, -g)|_inf for one camera-side block.
KBA_HD double block_grad_inf(int kind, const double* x, const double* g) {
    double out[7], d[6], m = 0.0;
    if (kind == 0) {
        for (int i = 0; i < 6; ++i) d[i] = -g[i];
        pose_plus(x, d, out);
        for (int i = 0; i < 7; ++i) m = fmax(m, fabs(x[i] - out[i]));
    } else if (kind == 1) {
        for (int i = 0; i < 3; ++i) d[i] = -g[i];
        unitvec_plus(x, d, out);
        for (int i = 0; i < 3; ++i) m = fmax(m, fabs(x[i] - out[i]));
    } else {
        m = fabs(g[0]);
    }
    return m;
}

// N per-lane values reduced over the workgroup, the first NSUM by addition, the rest by maximum; every lane gets the
// results.  red: N * nt doubles (host tree) / N * (nt / 64) doubles (device).  On gfx950 the waves reduce with
// butterfly shuffles (every lane of a wave ends with the same bits) and meet once in LDS: one barrier instead of
// log2(nt) of them - the trees were 4-5 us of each window-level kernel.
template <int N, int NSUM>
KBA_HD void coop_reduce(double* v, int tid, int nt, double* red) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nt >= 64 && (nt & 63) == 0) {
        const int wave = tid >> 6, nw = nt >> 6;
        for (int q = 0; q < N; ++q) {
            double x = v[q];
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(x, off, 64);
                x = q < NSUM ? x + o : fmax(x, o);
            }
            if ((tid & 63) == 0) red[wave * N + q] = x;
        }
        KBA_SYNC();
        for (int q = 0; q < N; ++q) {
            double x = red[q];
            for (int u = 1; u < nw; ++u) x = q < NSUM ? x + red[u * N + q] : fmax(x, red[u * N + q]);
            v[q] = x;
        }
        KBA_SYNC();
        return;
    }
#endif
    for (int q = 0; q < N; ++q) red[q * nt + tid] = v[q];
    KBA_SYNC();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            for (int q = 0; q < NSUM; ++q) red[q * nt + tid] += red[q * nt + tid + s];
            for (int q = NSUM; q < N; ++q) red[q * nt + tid] = fmax(red[q * nt + tid], red[q * nt + tid + s]);
        }
        KBA_SYNC();
    }
    for (int q = 0; q < N; ++q) v[q] = red[q * nt];
    KBA_SYNC();
}

// doubles coop_reduce<N, .> needs in `red` for nt lanes: one slot per wave on the device, a full tree on the host
KBA_HD int coop_red_doubles(int n, int nt) { return n * ((nt >= 64 && (nt & 63) == 0) ? nt >> 6 : nt); }

// scratch doubles needed by cam_assemble / cam_solve for a system of nc slots and nt lanes
constexpr int kGpChunk = 128;  // ground-plane rows staged in LDS per pass of cam_assemble
// A regulariser row touches at most two neighbouring keyframes: kept dense over their 2 x kCamSlots columns as
// [r | first keyframe | 20 values] so that the lane owning an entry of H can add the rows in row order.
constexpr int kRegDense = 2 + 2 * kCamSlots;
KBA_HD int cam_max_reg_rows(int nc) {  // upper bound of reg_row_count for a window of nc camera slots
    const int nkf = nc / kCamSlots;
    return 1 + (nkf > 0 ? (nkf - 1) * 5 + 3 * nkf : 0) + 3;
}
// scratch of cam_assemble: H (nc x nc) | regulariser rows, aliased by the ground-plane staging | dense regulariser
// rows, aliased by the views' raw camera-side totals of the first phase and by the reduction tree of the last
KBA_HD int cam_assemble_union(int nc) {
    const int rows = (int)((sizeof(RegRow) * cam_max_reg_rows(nc) + 7) / 8);
    const int gp = kGpChunk * 12;
    return rows > gp ? rows : gp;
}
// n_view: views of the window (their raw camera-side totals, n_view x kLinPartial doubles, share the third region)
KBA_HD int cam_assemble_scratch(int nc, int nt, int n_view) {
    const int dense = cam_max_reg_rows(nc) * kRegDense;
    const int red = coop_red_doubles(6, nt);
    const int raw = n_view * kLinPartial;
    const int third = red > dense ? red : dense;
    return nc * nc + cam_assemble_union(nc) + (raw > third ? raw : third);
}
// Windows whose scratch exceeds this many bytes work in global memory (WinDesc::cam_scr_off) instead of LDS (160 KB / CU).
constexpr int kCamLdsCapBytes = 150 * 1024;
KBA_HD int cam_solve_scratch(int nc, int nt, int nf = -1) {
    if (nf < 0) nf = nc;  // (nf: free slots of the compact system, <= nc)
    return nf * (nf + 1) + nf + nc + (nf + 1) / 2 + 1 + coop_red_doubles(3, nt) + nf * (nf + 1) / 2 + nf;  // A | y | dl | fl | red | Hs
}

// Workgroup-per-window: assemble the camera-camera normal equations H_cc, g_c at the linearisation point, the
// Jacobi scaling of the camera columns, cost / gradient-norm / |x| reductions.
// scratch: see cam_assemble_scratch.
KBA_HD void cam_assemble(const BatchView& bv, const SolveConsts& c, int w, int tid, int nt, double* scratch) {
    const WinDesc& wd = bv.win[w];
    const int nc = wd.nc;
    double* H = scratch;
    double* gps = scratch + nc * nc;  // ground-plane staging, reused by the regulariser rows afterwards
    RegRow* rows = reinterpret_cast<RegRow*>(gps);
    double* dense = gps + cam_assemble_union(nc);  // dense regulariser rows, reused by the reductions afterwards
    double* red = dense;
    double* gc = bv.gc + (int64_t)wd.cam0;
    KBA_TICK(0);
    for (int i = tid; i < nc * nc; i += nt) H[i] = 0.0;
    for (int i = tid; i < nc; i += nt) gc[i] = 0.0;
    // Ground-plane rows are staged through LDS in chunks (coalesced plane reads).  The first chunk is fetched here, ahead of
    // the observation sums: its memory round trip runs under theirs instead of after them.
    auto gp_stage = [&](int g0) {
        const int ng = (wd.gp0 + wd.n_gp - g0) < kGpChunk ? (wd.gp0 + wd.n_gp - g0) : kGpChunk;
        for (int i = tid; i < ng * 12; i += nt) {
            const int q = i / ng, g = g0 + i % ng;  // q-major: consecutive lanes read consecutive rows of a plane
            double v;
            if (q < 10)
                v = bv.gp_F[q * bv.SG + g];
            else if (q == 10)
                v = bv.gp_r[g];
            else
                v = (double)(bv.gp_kf[g] - wd.kf0);
            gps[(i % ng) * 12 + q] = v;
        }
    };
    if (wd.n_gp > 0 && !bv.gp_red) gp_stage(wd.gp0);
    KBA_SYNC();
    // (1) observations: the camera-side RAW sums of every view (lin_cam_half0 / _half1) totalled over the window's landmark
    //     workgroups - one lane per (view, entry), in workgroup order - then U_k (6x6) and g_k per keyframe from its views' totals
    //     (cam_raw_entry: the view's Rc enters here, once per window instead of once per pair); one lane per (keyframe, entry):
    //     single writer, fixed order.
    double* raw = dense;  // [n_view][kLinPartial]: the dense regulariser rows only move in behind phase (2)
    for (int e = tid; e < wd.n_view * kLinPartial; e += nt) {
        const double* lp = bv.lv_part + wd.lvpart_off + e;
        const int64_t stride = (int64_t)wd.n_view * kLinPartial;
        double acc = 0.0;
        int b = 0;
        for (; b + 4 <= wd.n_lblk; b += 4) {  // four loads in flight, added in workgroup order
            const double v0 = lp[b * stride], v1 = lp[(b + 1) * stride], v2 = lp[(b + 2) * stride], v3 = lp[(b + 3) * stride];
            acc += v0;
            acc += v1;
            acc += v2;
            acc += v3;
        }
        for (; b < wd.n_lblk; ++b) acc += lp[b * stride];
        raw[e] = acc;
    }
    KBA_SYNC();
    for (int e = tid; e < wd.n_kf * 27; e += nt) {
        const int kl = e / 27, q = e % 27;
        double acc = 0.0;
        for (int j = 0; j < wd.n_view; ++j) {
            if (bv.view_kf[wd.view0 + j] - wd.kf0 != kl) continue;
            acc += cam_raw_entry(q, raw + j * kLinPartial, bv.view_cam + 16 * (int64_t)(wd.view0 + j) + 4);
        }
        if (q < 21) {
            int a = 0, rem = q;
            while (rem >= 6 - a) {
                rem -= 6 - a;
                ++a;
            }
            const int bb = a + rem;
            H[(kl * kCamSlots + a) * nc + kl * kCamSlots + bb] += acc;
            if (bb != a) H[(kl * kCamSlots + bb) * nc + kl * kCamSlots + a] += acc;
        } else {
            gc[kl * kCamSlots + (q - 21)] += acc;
        }
    }
    KBA_SYNC();
    KBA_TICK(1);
    if (c.pad == 41) return;  // (41-44: profiling aids, early exits after the phases)
    // (2) ground-plane rows: F^T F on the 10x10 block of their keyframe: one lane per (keyframe, entry) adds the staged
    //     rows of ITS keyframe in row order.
    if (bv.gp_red) {
        // landmark-sharded solve: every shard has folded ITS rows into F^T F | F^T r per keyframe (shard_reduce_lin); here the
        // P contributions are added in shard order (wd.gp0 / n_gp of the consumer view describe "one row per shard")
        for (int e = tid; e < wd.n_kf * 110; e += nt) {
            const int kl = e / 110, q = e % 110;
            const int a = q < 100 ? q / 10 : q - 100, bb = q < 100 ? q % 10 : 10;
            const int lo = bb == 10 ? a : (a < bb ? a : bb), hi = bb == 10 ? 10 : (a < bb ? bb : a);
            const int idx = hi == 10 ? 55 + lo : lo * 10 - lo * (lo - 1) / 2 + (hi - lo);
            double acc = 0.0;
            for (int sh = 0; sh < bv.gp_red_P; ++sh) acc += bv.gp_red[((int64_t)sh * bv.TK + wd.kf0 + kl) * kGpRed + idx];
            if (bb < 10)
                H[(kl * kCamSlots + a) * nc + kl * kCamSlots + bb] += acc;
            else
                gc[kl * kCamSlots + a] += acc;
        }
        KBA_SYNC();
    }
    for (int g0 = wd.gp0; !bv.gp_red && g0 < wd.gp0 + wd.n_gp; g0 += kGpChunk) {
        const int ng = (wd.gp0 + wd.n_gp - g0) < kGpChunk ? (wd.gp0 + wd.n_gp - g0) : kGpChunk;
        if (g0 != wd.gp0) {  // (the first chunk is in LDS already)
            gp_stage(g0);
            KBA_SYNC();
        }
#if defined(__HIP_DEVICE_COMPILE__)
        // gfx950, 256 lanes (round 5): [F | r]^T [F | r] of a keyframe's rows is ONE 16 x 16 Gram tile - wave kl (mod 4) runs
        // v_mfma_f64_16x16x4 over the keyframe's staged rows, four rows per instruction (operand lane (i, k) = entry i of row
        // 4 s + k, zero beyond the row's 11 values and beyond the keyframe's rows), and adds the tile's entries (a, b < 10) to H,
        // (a, 10) to g_c.  ~100 rows per keyframe at C2: 25 dependent MFMAs instead of ~100 dependent multiply-adds behind two LDS
        // reads each, on three passes of the workgroup (11 of the 30 us of this function, profiles/r03_single_window_phase_ticks.txt).
        // Same products; an entry's rows are added four at a time inside the instruction (all device paths share this code, the
        // CPU-tier emulation keeps the loop below).
        if (nt == 256) {
            typedef double v4d_t __attribute__((ext_vector_type(4)));
            const int lane = tid & 63, li = lane & 15, kq = lane >> 4;
            for (int kl = tid >> 6; kl < wd.n_kf; kl += 4) {
                int lo = bv.kf_gp0[wd.kf0 + kl] - g0, hi = lo + bv.kf_ngp[wd.kf0 + kl];
                lo = lo < 0 ? 0 : lo;
                hi = hi > ng ? ng : hi;
                if (hi <= lo) continue;  // (uniform over the wave)
                v4d_t acc = {0.0, 0.0, 0.0, 0.0};
                for (int g = lo; g < hi; g += 8) {  // two instructions per pass: their operand reads are in flight together
                    const int r0 = g + kq, r1 = g + 4 + kq;
                    const double z0 = (r0 < hi && li < 11) ? gps[r0 * 12 + li] : 0.0;
                    const double z1 = (r1 < hi && li < 11) ? gps[r1 * 12 + li] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z1, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // f64 16x16x4 C / D layout: row = (lane >> 4) + 4 r, column = lane & 15
                    const int a = kq + 4 * r;
                    if (a < 10 && li < 10)
                        H[(kl * kCamSlots + a) * nc + kl * kCamSlots + li] += acc[r];
                    else if (a < 10 && li == 10)
                        gc[kl * kCamSlots + a] += acc[r];
                }
            }
        } else
#endif
        for (int e = tid; e < wd.n_kf * 110; e += nt) {
            const int kl = e / 110, q = e % 110;
            const int a = q < 100 ? q / 10 : q - 100, bb = q < 100 ? q % 10 : 10;
            // rows are sorted by keyframe: this keyframe owns [kf_gp0, kf_gp0 + kf_ngp), clipped to the chunk
            int lo = bv.kf_gp0[wd.kf0 + kl] - g0, hi = lo + bv.kf_ngp[wd.kf0 + kl];
            lo = lo < 0 ? 0 : lo;
            hi = hi > ng ? ng : hi;
            double acc = 0.0;
            for (int g = lo; g < hi; ++g) acc += gps[g * 12 + a] * gps[g * 12 + bb];
            if (bb < 10)
                H[(kl * kCamSlots + a) * nc + kl * kCamSlots + bb] += acc;
            else
                gc[kl * kCamSlots + a] += acc;
        }
        KBA_SYNC();
    }
    KBA_TICK(2);
    if (c.pad == 42) return;
    // (3) regulariser rows: one lane evaluates one row into scratch and spreads it over the columns of its (at most
    //     two, neighbouring) keyframes ...
    const int nrows = reg_row_count(wd);
    // Row -> lane: the row kinds differ a lot (pose-pair rows differentiate a relative translation, the others are
    // a few multiplications), and lanes of one wave that take different kinds run them one after the other.  With a
    // full workgroup every wave gets one kind: 0 motion rows, 1 normal-difference rows, 2 distance rows (+ the
    // scale row), 3 the per-keyframe and speed rows.
    const int reg_base = wd.has_scale_reg ? 1 : 0, reg_npair = wd.has_gp_reg ? wd.n_kf - 1 : 0;
    for (int i0 = tid; i0 < (nt >= 256 ? nt : nrows); i0 += nt) {
        int i = i0;
        if (nt >= 256) {
            const int wv = tid >> 6, ln = tid & 63;
            i = -1;
            if (wv == 0) {
                if (ln < reg_npair) i = reg_base + ln * 5 + 4;
            } else if (wv == 1) {
                if (ln < reg_npair * 3) i = reg_base + (ln / 3) * 5 + ln % 3;
            } else if (wv == 2) {
                if (ln < reg_npair)
                    i = reg_base + ln * 5 + 3;
                else if (ln == 63 && wd.has_scale_reg)
                    i = 0;
            } else if (wv == 3) {
                if (reg_base + reg_npair * 5 + ln < nrows) i = reg_base + reg_npair * 5 + ln;
            }
            if (i < 0) continue;
        }
        int all_const;
        reg_row_eval(wd, bv.cmask, bv.pose, bv.pdir, bv.pdist, i, true, rows[i], all_const);
        if (all_const) rows[i].n = -1;  // fixed-cost row
        const RegRow& row = rows[i];
        double* dd = dense + i * kRegDense;
        int klo = 1 << 20;
        for (int p = 0; p < row.n; ++p) klo = row.col[p] / kCamSlots < klo ? row.col[p] / kCamSlots : klo;
        if (row.n <= 0) klo = -(1 << 20);  // contributes nowhere
        for (int q = 0; q < 2 * kCamSlots; ++q) dd[2 + q] = 0.0;
        for (int p = 0; p < row.n; ++p) dd[2 + row.col[p] - klo * kCamSlots] = row.val[p];
        dd[0] = row.r;
        dd[1] = (double)klo;
    }
    KBA_SYNC();
    KBA_TICK(3);
    if (c.pad == 45) return;
    // ... then every entry of the block tridiagonal of H (diagonal blocks, then the blocks (k, k+1) and their mirror
    //     images) is owned by one lane, which adds the rows' products in row order: no serial pass over the rows,
    //     same sum per entry for any lane count.
    for (int e = tid; e < (2 * wd.n_kf - 1) * kCamSlots * kCamSlots; e += nt) {
        const int blk = e / (kCamSlots * kCamSlots), q = e % (kCamSlots * kCamSlots);
        const int ka = blk < wd.n_kf ? blk : blk - wd.n_kf, kb = blk < wd.n_kf ? blk : ka + 1;
        const int a = ka * kCamSlots + q / kCamSlots, b = kb * kCamSlots + q % kCamSlots;
        double acc = H[a * nc + b];
        reg_rows_of_block(wd, ka, kb, [&](int lo, int cnt, int base) {
            const double* da = dense + lo * kRegDense + 2 + a - base * kCamSlots;
            const double* db = dense + lo * kRegDense + 2 + b - base * kCamSlots;
            for (int i = 0; i < cnt; ++i) acc += da[i * kRegDense] * db[i * kRegDense];
        });
        H[a * nc + b] = acc;
        if (ka != kb) H[b * nc + a] = acc;
    }
    for (int a = tid; a < nc; a += nt) {
        double acc = gc[a];
        reg_rows_of_block(wd, a / kCamSlots, a / kCamSlots, [&](int lo, int cnt, int base) {
            const double* da = dense + lo * kRegDense + 2 + a - base * kCamSlots;
            const double* dr = dense + lo * kRegDense;
            for (int i = 0; i < cnt; ++i) acc += da[i * kRegDense] * dr[i * kRegDense];
        });
        gc[a] = acc;
    }
    KBA_SYNC();
    KBA_TICK(4);
    if (c.pad == 43) return;
    // (4) mask constant / absent slots
    const uint8_t* cm = bv.cmask + (int64_t)wd.cam0;
    for (int i = tid; i < nc * nc; i += nt) {
        if (!cm[i / nc] || !cm[i % nc]) H[i] = 0.0;
    }
    for (int i = tid; i < nc; i += nt)
        if (!cm[i]) gc[i] = 0.0;
    KBA_SYNC();
    double* Hg = bv.Hcc + wd.hcc_off;
    for (int i = tid; i < nc * nc; i += nt) Hg[i] = H[i];
    if (bv.st[w].compute_scale) {
        for (int i = tid; i < nc; i += nt)
            bv.scale_c[wd.cam0 + i] = (cm[i] && c.jacobi_scaling) ? 1.0 / (1.0 + sqrt(H[i * nc + i])) : 1.0;
    }
    KBA_TICK(5);
    if (c.pad == 44) return;
    // (5) reductions
    double cost = 0.0, failf = 0.0, gmax = 0.0, xn2 = 0.0, reg_free = 0.0, reg_fixed = 0.0;
    for (int i = tid; i < wd.n_lblk * wd.n_view; i += nt) cost += bv.lv_part[wd.lvpart_off + (int64_t)i * kLinPartial];
    for (int b = wd.lblk0 + tid; b < wd.lblk0 + wd.n_lblk; b += nt)
        if (bv.lblk_linfail[b] != 0.0) failf = 1.0;
    for (int g = wd.gp0 + tid; g < wd.gp0 + wd.n_gp; g += nt) cost += bv.gp_cost[g];
    for (int i = tid; i < nrows; i += nt) {
        if (rows[i].n < 0)
            reg_fixed += 0.5 * rows[i].r * rows[i].r;
        else
            reg_free += 0.5 * rows[i].r * rows[i].r;
    }
    for (int b = wd.lblk0 + tid; b < wd.lblk0 + wd.n_lblk; b += nt) {
        gmax = fmax(gmax, bv.lblk_part[(int64_t)b * 8 + 0]);
        xn2 += bv.lblk_part[(int64_t)b * 8 + 1];
    }
    for (int k = tid; k < wd.n_kf; k += nt) {
        const int gk = wd.kf0 + k;
        if (cm[k * kCamSlots + 0]) {
            const double* x = bv.pose + 7 * (int64_t)gk;
            gmax = fmax(gmax, block_grad_inf(0, x, gc + k * kCamSlots));
            for (int i = 0; i < 7; ++i) xn2 += x[i] * x[i];
        }
        if (cm[k * kCamSlots + 6]) {
            const double* x = bv.pdir + 3 * (int64_t)gk;
            gmax = fmax(gmax, block_grad_inf(1, x, gc + k * kCamSlots + 6));
            for (int i = 0; i < 3; ++i) xn2 += x[i] * x[i];
        }
        if (cm[k * kCamSlots + 9]) {
            gmax = fmax(gmax, block_grad_inf(2, bv.pdist + gk, gc + k * kCamSlots + 9));
            xn2 += bv.pdist[gk] * bv.pdist[gk];
        }
    }
    {   // six reductions at once (4 sums, 2 maxima); red holds 6*nt doubles
        double v[6] = {reg_free, reg_fixed, cost, xn2, failf, gmax};
        coop_reduce<6, 4>(v, tid, nt, red);
        reg_free = v[0];
        reg_fixed = v[1];
        cost = v[2] + reg_free;
        xn2 = v[3];
        failf = v[4];
        gmax = v[5];
    }
    if (tid == 0) {
        WinRed& r = bv.red[w];
        r.lin_cost = cost;
        r.lin_fail = failf != 0.0;
        r.gmax = gmax;
        r.xnorm2 = xn2;
        bv.reg_cost[2 * w] = reg_free;
        bv.reg_cost[2 * w + 1] = reg_fixed;
    }
    KBA_TICK(6);
}

// Workgroup-per-window: S = S_c H S_c + D^2 - sum Schur slabs, rhs = S_c g_c - sum slabs; dense Cholesky; camera
// step, camera candidate, camera parts of the step reductions.  scratch: S (nc*nc) | v (3*nc) | red (nt).
KBA_HD void cam_solve(const BatchView& bv, const SolveConsts& c, int w, int tid, int nt, double* scratch, int* flag) {
    // Works on the COMPACT system of the nf free slots.  scratch: A (nf x (nf+1), the rhs is column nf) | y (nf) |
    // dl (nc) | fl (nf ints) | red (nt).
    const WinDesc& wd = bv.win[w];
    const int nc = wd.nc, nf = wd.nf, nfp = wd.nf_pad, lda = nf + 1;
    double* A = scratch;
    double* y = A + nf * lda;
    double* dl = y + nf;
    int* fl = reinterpret_cast<int*>(dl + nc);
    double* red = dl + nc + (nf + 1) / 2 + 1;
    // Hs: the entries of the scaled, UNDAMPED camera system and rhs (S_c H S_c | S_c g_c; upper triangle + rhs in the order of the
    // assembly below) - what the model cost change of the step needs, kept from the assembly instead of read again from memory
    double* Hs = red + coop_red_doubles(3, nt);
    const double radius = bv.st[w].radius;
    const double* Hg = bv.Hcc + wd.hcc_off;
    const double* sc = bv.scale_c + wd.cam0;
    const int32_t* cs = bv.cslot + wd.cam0;
    double* yc = bv.yc + wd.cam0;
    double* dc = bv.delta_c + wd.cam0;
    const int slab = c.schur_packed ? schur_need_pad(nf) : nfp * nfp;
    const int nfq = wd.nfq;
    KBA_TICK(8);
    for (int a = tid; a < nc; a += nt)
        if (cs[a] >= 0) fl[cs[a]] = a;
    KBA_SYNC();
    // ---- assemble [S | rhs]: S = S_c H S_c + D^2 - sum slabs (upper triangle), rhs = S_c g_c - sum slabs
    // partial slabs of the Schur workgroups, or (landmark-sharded solve) the per-shard sums of them
    const double* sp = c.schur_nslab > 0 ? bv.S_red + wd.sred_off : bv.S_part + wd.spart_off;
    // Entries of the upper triangle + rhs column, enumerated row by row (row ca: cb = ca..nf) so that every lane gets
    // the same share; a lane sums up to 4 entries at once, 4 slabs each: 16 independent loads in flight (one window
    // alone on the GPU is bound by exactly this latency chain).  Per entry the order of the sum stays q mod 4.
    const int n_need = nf * (nf + 1) / 2 + nf;
    const int n_slab = c.schur_nslab > 0 ? c.schur_nslab : schur_slabs(wd, c.schur_span, c.schur_span_gp);
    // The slabs of the PLAIN groups (landmarks without a ground-plane row) come first and are zero outside the pose slots and
    // the rhs: an entry that involves a plane slot skips them - from a multiple of four on, so that every slab keeps its place
    // in the (q mod 4) order and the sums keep their bits (a skipped term is an exact zero).  Round 5: this sum is what bounds
    // k_cam_solve in a batch (137 KB of slab entries per window and iteration at C2; 86 KB with the skip).
    const int q_gp = c.schur_nslab > 0 ? 0 : (schur_plain_slabs(wd, c.schur_span) & ~3);
    constexpr int kE = KBA_SLAB_ENTRIES;  // entries a lane sums at once
    for (int i0 = tid; i0 < n_need; i0 += kE * nt) {
        double s[kE], acc[kE][4];
        int64_t off[kE];
        int dst[kE], qs[kE];
        for (int e = 0; e < kE; ++e) {
            const int i = i0 + e * nt;
            dst[e] = -1;
            off[e] = 0;
            qs[e] = n_slab;
            s[e] = 0.0;
            for (int r = 0; r < 4; ++r) acc[e][r] = 0.0;
            if (i >= n_need) continue;
            int ca, cb;
            schur_need_decode(i, nf, ca, cb);
            const int a = fl[ca];
            if (cb < nf) {
                const int b = fl[cb];
                double v = sc[a] * sc[b] * Hg[a * nc + b];
                Hs[i] = v;
                if (ca == cb) v += fmin(fmax(v, c.min_lm_diagonal), c.max_lm_diagonal) / radius;
                s[e] = v;
            } else {
                s[e] = sc[a] * bv.gc[wd.cam0 + a];
                Hs[i] = s[e];
            }
            // (packed slabs of a sharded solve hold exactly these entries in this order)
            off[e] = c.schur_packed ? (int64_t)i : schur_need_offset(ca, cb, nf, nfq, nfp);
            dst[e] = ca * lda + cb;
            qs[e] = (ca >= nfq || (cb >= nfq && cb < nf)) ? q_gp : 0;
        }
        // (branch-free: a skipped term loads the same entry of the LAST slab instead - a line the sum reads anyway - and adds an exact
        // zero; with a branch around the loads the 16 loads of a step were no longer in flight together, and a window with 79 slabs -
        // C4 - took 30 % longer per LM iteration)
        int q = 0;
        for (; q + 4 <= n_slab; q += 4)
            for (int e = 0; e < kE; ++e)
                for (int r = 0; r < 4; ++r) {
                    const bool on = q >= qs[e];
                    const double v = sp[(int64_t)(on ? q + r : n_slab - 1) * slab + off[e]];
                    acc[e][r] += on ? v : 0.0;
                }
        for (; q < n_slab; ++q)
            for (int e = 0; e < kE; ++e) {
                const bool on = q >= qs[e];
                const double v = sp[(int64_t)(on ? q : n_slab - 1) * slab + off[e]];
                acc[e][0] += on ? v : 0.0;
            }
        for (int e = 0; e < kE; ++e)
            if (dst[e] >= 0) A[dst[e]] = s[e] - ((acc[e][0] + acc[e][1]) + (acc[e][2] + acc[e][3]));
    }
    KBA_SYNC();
    KBA_TICK(9);
    if (c.pad == 1) return;
    // ---- right-looking Cholesky of the upper triangle fused with the forward substitution (rhs = extra column):
    //      A = U^T U, y = U^-T rhs.  One barrier per pivot; the rows are divided by sqrt(d_k) in one pass at the end.
    //      Eigen LLT<Upper> semantics: failure when a pivot is <= 0.
    //      The kernel's time is the slowest wave's path through the pivots (cycle counters: the first wave of a
    //      workgroup used to own the four longest rows of every trailing update, all of the previous row's sqrt +
    //      division chain and the diagonal's sqrt behind the barrier: 1830 cycles per pivot against 800 for the
    //      last wave), so trailing rows go round-robin over the waves and nothing but the update sits between barriers.
    bool failed = false;
    (void)flag;
    const int tw = nt >= 16 ? 16 : 1, th = nt / tw, tx = tid % tw;
    const int ty = nt == 256 ? ((tid >> 4) & 3) * 4 + (tid >> 6) : tid / tw;  // wave w: rows w, w + 4, w + 8, w + 12 (+16 ...)
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950, 256 lanes: the same elimination in blocks of FOUR pivots (round 4).  The chain of one pivot - read the pivot, divide,
    // update, barrier - is ~1150 cycles however the lanes are dealt (rounds 2-3), 40 of them in a row at C2.  Here ONE wave runs
    // the four pivots of a block on the block's own (<= 3) remaining rows, ordered by wave-level fences only; then all four waves
    // apply the block to the trailing rows as a rank-4 update, A[i][j] -= sum_k (A[k][i] / d_k) A[k][j], one 16 x 16 tile per
    // v_mfma_f64_16x16x4 (A operand: the negated multipliers, B operand: the pivot rows, C: the tile, all straight from LDS).
    // Two workgroup barriers per block instead of four.  Same pivots, same failure rule; the four products of an entry are added
    // inside the MFMA instead of one after the other, so results differ from the loop below by rounding (all device paths share
    // this code; the CPU-tier emulation keeps the loop).
    if (nt == 256) {
        typedef double v4d_t __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
        double* invd = y;  // (y is free until the factorisation is over)
        if (tid == 0) *flag = 0;
        KBA_SYNC();
        for (int kb = 0; kb < nf; kb += 4) {
            const int kbe = kb + 4 < nf ? kb + 4 : nf;
            if (wave == 0 && nf < 64) {
                // the block's four rows in registers, lane = column: pivots and multipliers travel by v_readlane, no memory round
                // trip between the four pivots (through LDS the block's own rows cost ~850 cycles per pivot - no gain over the loop)
                auto bcast = [](double v, int from) {
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), from), __builtin_amdgcn_readlane(__double2loint(v), from));
                };
                double rw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rw[q] = (kb + q < kbe && lane <= nf) ? A[(kb + q) * lda + lane] : 0.0;
                bool bad = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (kb + q < kbe && !bad) {
                        const double d = bcast(rw[q], kb + q);
                        if (!(d > 0.0)) {  // (uniform over the wave)
                            bad = true;
                        } else {
                            const double inv_d = 1.0 / d;
                            if (lane == 0) invd[q] = inv_d;
#pragma unroll
                            for (int pr = q + 1; pr < 4; ++pr) {
                                if (kb + pr < kbe) {
                                    const double aki = bcast(rw[q], kb + pr) * inv_d;
                                    if (lane >= kb + pr) rw[pr] -= aki * rw[q];
                                }
                            }
                        }
                    }
                }
                if (bad && lane == 0) *flag = 1;
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (kb + q < kbe && lane >= kb + q && lane <= nf) A[(kb + q) * lda + lane] = rw[q];
            } else if (wave == 0) {
                for (int k = kb; k < kbe; ++k) {
                    const double d = A[k * lda + k];
                    if (!(d > 0.0)) {  // (uniform over the wave)
                        if (lane == 0) *flag = 1;
                        break;
                    }
                    const double inv_d = 1.0 / d;
                    if (lane == 0) invd[k - kb] = inv_d;
                    for (int i = k + 1; i < kbe; ++i) {
                        const double aki = A[k * lda + i] * inv_d;
                        for (int j = i + lane; j <= nf; j += 64) A[i * lda + j] -= aki * A[k * lda + j];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
            }
            KBA_SYNC();
            if (*flag) break;  // (uniform over the workgroup)
            const int n_tr = nf - kbe;  // trailing rows kbe .. nf - 1; trailing columns kbe .. nf (the rhs rides along)
            if (n_tr > 0) {
                const int Ti = (n_tr + 15) / 16, Tj = (n_tr + 1 + 15) / 16;
                const int kp = kb + kq;  // this lane's pivot of the block (operand index k of the MFMA)
                const bool kin = kp < kbe;
                const double ivd = kin ? invd[kq] : 0.0;
                int t = 0;
                for (int ti = 0; ti < Ti; ++ti)
                    for (int tj = ti; tj < Tj; ++tj, ++t) {
                        if ((t & 3) != wave) continue;
                        const int i_op = kbe + 16 * ti + li, j_op = kbe + 16 * tj + li;
                        const double wv = kin && i_op < nf ? -(A[kp * lda + i_op] * ivd) : 0.0;
                        const double pv = kin && j_op <= nf ? A[kp * lda + j_op] : 0.0;
                        v4d_t cacc;
                        const int jc = kbe + 16 * tj + li;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ic = kbe + 16 * ti + kq + 4 * r;
                            cacc[r] = (ic < nf && jc <= nf && jc >= ic) ? A[ic * lda + jc] : 0.0;
                        }
                        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, pv, cacc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ic = kbe + 16 * ti + kq + 4 * r;
                            if (ic < nf && jc <= nf && jc >= ic) A[ic * lda + jc] = cacc[r];
                        }
                    }
            }
            KBA_SYNC();
        }
        failed = *flag != 0;
    } else
#endif
    for (int k = 0; k < nf; ++k) {
        const double d = A[k * lda + k];
        if (!(d > 0.0)) {
            failed = true;  // uniform: every lane reads the same pivot
            break;
        }
        const double inv_d = 1.0 / d;
        for (int i = k + 1 + ty; i < nf; i += th) {  // 2-D lane grid: no integer division in the inner loop
            const double aki = A[k * lda + i] * inv_d;
            for (int j = i + tx; j <= nf; j += tw) A[i * lda + j] -= aki * A[k * lda + j];
        }
        KBA_SYNC();
    }
    if (failed) {
        if (tid == 0) {
            WinRed& r = bv.red[w];
            r.chol_fail = 1;
            r.mcc = 0.0;
            r.step2 = 0.0;
            r.cand2 = 0.0;
        }
        for (int a = tid; a < nc; a += nt) dc[a] = 0.0;
        return;
    }
    // U[k][j] = A[k][j] / sqrt(d_k), U[k][k] = sqrt(d_k)   (y holds the square roots until the substitution below)
    for (int k = tid; k < nf; k += nt) y[k] = sqrt(A[k * lda + k]);
    KBA_SYNC();
    for (int k = ty; k < nf; k += th) {
        const double dp = y[k];
        for (int j = k + tx; j <= nf; j += tw) A[k * lda + j] = j == k ? dp : A[k * lda + j] / dp;
    }
    KBA_SYNC();
    KBA_TICK(10);
    if (c.pad == 2) return;
    // ---- backward substitution U x = y (y = column nf), column oriented
#if defined(__HIP_DEVICE_COMPILE__)
    if (nt >= 64 && nf <= 64) {
        // gfx950: by the first wave alone, y_j in a register of lane j, x_i broadcast by v_readlane - no barrier and no
        // LDS round trip on the chain of nf dependent unknowns (same operations per entry as the loop below)
        if (tid < 64) {
            const bool mine = tid < nf;
            double yv = mine ? A[tid * lda + nf] : 0.0;
            const double diag = mine ? A[tid * lda + tid] : 1.0;
            for (int i = nf - 1; i >= 0; --i) {
                const double xi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(yv), i), __builtin_amdgcn_readlane(__double2loint(yv), i)) /
                                  __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(diag), i), __builtin_amdgcn_readlane(__double2loint(diag), i));
                if (tid < i)
                    yv -= A[tid * lda + i] * xi;
                else if (tid == i)
                    yv = xi;
            }
            if (mine) y[tid] = yv;
        }
    } else
#endif
    {   // one barrier per unknown
        for (int i = tid; i < nf; i += nt) y[i] = A[i * lda + nf];
        KBA_SYNC();
        for (int i = nf - 1; i >= 0; --i) {
            const double xi = y[i] / A[i * lda + i];
            for (int j = tid; j < i; j += nt) y[j] -= A[j * lda + i] * xi;
            KBA_SYNC();
            if (tid == 0) y[i] = xi;
        }
    }
    KBA_SYNC();
    KBA_TICK(11);
    if (c.pad == 3) return;
    for (int a = tid; a < nc; a += nt) {
        const int ca = cs[a];
        const double yv = ca >= 0 ? y[ca] : 0.0;
        yc[a] = yv;
        const double d = ca >= 0 ? -sc[a] * yv : 0.0;
        dc[a] = d;
        dl[a] = d;
    }
    KBA_SYNC();
    // camera part of the model cost change: -g_c.d - 1/2 d^T H_cc d (unscaled) = g'.y - 1/2 y^T H' y in the scaled variables of the
    // solve (d = -S_c y, g' = S_c g_c, H' = S_c H_cc S_c), from the entries the assembly kept in LDS (Hs): every lane takes the
    // entries it assembled.  (Rounds 2-5 read the nc columns of H_cc from memory again here: ten dependent round trips per lane.)
    double part = 0.0;
    for (int i = tid; i < n_need; i += nt) {
        int ca, cb;
        schur_need_decode(i, nf, ca, cb);
        if (cb < nf)
            part -= (ca == cb ? 0.5 : 1.0) * (Hs[i] * y[ca] * y[cb]);
        else
            part += Hs[i] * y[ca];
    }
    KBA_TICK(14);
    const uint8_t* cm = bv.cmask + (int64_t)wd.cam0;
    double step2 = 0.0, cand2 = 0.0;
    for (int k = tid; k < wd.n_kf; k += nt) {
        const int gk = wd.kf0 + k;
        const double* d = dl + k * kCamSlots;
        const double* x = bv.pose + 7 * (int64_t)gk;
        double* xc = bv.pose_c + 7 * (int64_t)gk;
        double pc[7];  // candidate pose (kept in registers for the per-view constants below)
        if (cm[k * kCamSlots + 0]) {
            pose_plus(x, d, pc);
            for (int i = 0; i < 7; ++i) {
                step2 += (x[i] - pc[i]) * (x[i] - pc[i]);
                cand2 += pc[i] * pc[i];
            }
        } else {
            for (int i = 0; i < 7; ++i) pc[i] = x[i];
        }
        for (int i = 0; i < 7; ++i) xc[i] = pc[i];
        const double* n = bv.pdir + 3 * (int64_t)gk;
        double* ncand = bv.pdir_c + 3 * (int64_t)gk;
        if (cm[k * kCamSlots + 6]) {
            unitvec_plus(n, d + 6, ncand);
            for (int i = 0; i < 3; ++i) {
                step2 += (n[i] - ncand[i]) * (n[i] - ncand[i]);
                cand2 += ncand[i] * ncand[i];
            }
        } else {
            for (int i = 0; i < 3; ++i) ncand[i] = n[i];
        }
        if (cm[k * kCamSlots + 9]) {
            bv.pdist_c[gk] = bv.pdist[gk] + d[9];
            step2 += d[9] * d[9];
            cand2 += bv.pdist_c[gk] * bv.pdist_c[gk];
        } else {
            bv.pdist_c[gk] = bv.pdist[gk];
        }
    }
    // ... and the per-view constants of the candidate poses (candidate cost, cf. view_consts_item): one lane per VIEW,
    // taken from the far end of the workgroup - other waves than the keyframe lanes above, so the two chains of dependent
    // memory round trips (pose -> candidate -> stores; view -> its keyframe's pose -> candidate -> constants) run side by
    // side instead of one keyframe lane walking over the window's views.  The candidate pose is formed again from the
    // same operands (same bits).
    for (int j = nt - 1 - tid; j < wd.n_view; j += nt) {
        const int view = wd.view0 + j, gk = bv.view_kf[view], k = gk - wd.kf0;
        const double* cam = bv.view_cam + 16 * (int64_t)view;
        const double* x = bv.pose + 7 * (int64_t)gk;
        double pc[7];
        if (cm[k * kCamSlots + 0]) {
            pose_plus(x, dl + k * kCamSlots, pc);
        } else {
            for (int i = 0; i < 7; ++i) pc[i] = x[i];
        }
        double Rk[9];
        quat_R(pc, Rk);
        double* vl = bv.view_lin_c + (int64_t)kViewLin * view;
        mat3_mul(cam + 4, Rk, vl);
        for (int i = 0; i < 3; ++i)
            vl[9 + i] = cam[4 + 3 * i] * pc[4] + cam[4 + 3 * i + 1] * pc[5] + cam[4 + 3 * i + 2] * pc[6] + cam[13 + i];
        vl[25] = cam[0];
        vl[26] = cam[1];
        vl[27] = cam[2];
        {   // what the back-substitution needs of the proposed camera step, per view: F_pose delta of an observation is
            // c^T Rc (dR p + delta_t) with dR = derivative of R(q) along the rotation step (kba_math.hpp:quat_dR):
            // K = Rc dR (9) and k0 = Rc delta_t (3); zeros for a keyframe whose pose block is not free
            const bool fr = cm[k * kCamSlots] != 0;
            const double* dk = dl + k * kCamSlots;
            const double dz[6] = {fr ? dk[0] : 0.0, fr ? dk[1] : 0.0, fr ? dk[2] : 0.0, fr ? dk[3] : 0.0, fr ? dk[4] : 0.0, fr ? dk[5] : 0.0};
            double dR[9];
            quat_dR(x, dz, dR);
            mat3_mul(cam + 4, dR, vl + 28);
            mat3_vec(cam + 4, dz + 3, vl + 37);
        }
    }
    KBA_TICK(12);
    // three sums at once (red holds 3*nt doubles)
    double v3[3] = {part, step2, cand2};
    coop_reduce<3, 3>(v3, tid, nt, red);
    if (tid == 0) {
        WinRed& r = bv.red[w];
        r.chol_fail = 0;
        r.mcc = v3[0];
        r.step2 = v3[1];
        r.cand2 = v3[2];
    }
    KBA_TICK(13);
}


// After backsub + candidate cost kernels: fold the landmark / observation partials into WinRed (workgroup, red[nt]).
// n_work: the lanes that take entries (the first n_work of the nt lanes that meet in the reduction; default all).  The
// lanes past n_work add zeros, so a 256-lane workgroup with n_work = 64 forms the sums of a 64-lane one (k_solve_wg).
KBA_HD void reduce_step(const BatchView& bv, int w, int tid, int nt, double* red, int n_work = -1) {
    const WinDesc& wd = bv.win[w];
    if (n_work < 0) n_work = nt;
    double mcc = 0.0, s2 = 0.0, c2 = 0.0, lfail = 0.0, cost = 0.0, cfail = 0.0;
    const int t0 = tid < n_work ? tid : (1 << 28);  // lanes past n_work: every loop below is empty
    for (int b = wd.lblk0 + t0; b < wd.lblk0 + wd.n_lblk; b += n_work) {
        mcc += bv.lblk_part[(int64_t)b * 8 + 2];
        s2 += bv.lblk_part[(int64_t)b * 8 + 3];
        c2 += bv.lblk_part[(int64_t)b * 8 + 4];
        if (bv.lblk_part[(int64_t)b * 8 + 5] != 0.0) lfail = 1.0;
    }
    for (int b = wd.lblk0 + t0; b < wd.lblk0 + wd.n_lblk; b += n_work) {  // candidate cost of the observations (backsub_lane)
        cost += bv.lblk_part[(int64_t)b * 8 + 6];
        if (bv.lblk_part[(int64_t)b * 8 + 7] != 0.0) cfail = 1.0;
    }
    {   // (four loads in flight, added in row order: one workgroup alone waits a memory round trip per load otherwise)
        int g = wd.gp0 + t0;
        const int g_end = wd.gp0 + wd.n_gp;
        for (; g + 3 * n_work < g_end; g += 4 * n_work) {
            const double v0 = bv.gp_cost_c[g], v1 = bv.gp_cost_c[g + n_work], v2 = bv.gp_cost_c[g + 2 * n_work], v3 = bv.gp_cost_c[g + 3 * n_work];
            cost += v0;
            cost += v1;
            cost += v2;
            cost += v3;
        }
        for (; g < g_end; g += n_work) cost += bv.gp_cost_c[g];
    }
    const int nrows = reg_row_count(wd);
    for (int i = t0; i < nrows; i += n_work) {
        RegRow row;
        int all_const;
        reg_row_eval(wd, bv.cmask, bv.pose_c, bv.pdir_c, bv.pdist_c, i, false, row, all_const);
        if (!all_const) cost += 0.5 * row.r * row.r;
    }
    double v[6] = {mcc, s2, c2, cost, lfail, cfail};
    coop_reduce<6, 4>(v, tid, nt, red);  // red: 6 * nt doubles on the host, 6 per wave on the device
    mcc = v[0];
    s2 = v[1];
    c2 = v[2];
    cost = v[3];
    lfail = v[4];
    cfail = v[5];
    if (tid == 0) {
        WinRed& r = bv.red[w];
        r.mcc += mcc;
        r.step2 += s2;
        r.cand2 += c2;
        if (lfail != 0.0) r.chol_fail = 1;
        r.cand_cost = cost;
        r.cand_fail = cfail != 0.0;
    }
}

// ======================================================================================= landmark sharding: a shard's contribution
// What shard `shard` of P puts into an exchange, folded over ITS landmark workgroups / ground-plane rows in their order
// (BatchView::x_*; the consumer adds the P contributions in shard order).  One workgroup (nt lanes) per window.
//   after the linearisation (before cam_assemble):  x_lv, x_lf, x_gp, x_gc, x_lb[0, 1, 5]
KBA_HD void shard_reduce_lin(const BatchView& bv, int w, int shard, int tid, int nt) {
    const WinDesc& wd = bv.win[w];
    const int n_e = wd.n_view * kLinPartial;
    for (int e = tid; e < n_e; e += nt) {
        double acc = 0.0;
        for (int b = 0; b < wd.n_lblk; ++b)
            if (bv.lblk_owner[wd.lblk0 + b] == shard) acc += bv.lv_part[wd.lvpart_off + (int64_t)b * n_e + e];
        bv.x_lv[wd.xlv_off + e] = acc;
    }
    for (int e = tid; e < wd.n_kf * kGpRed; e += nt) {
        const int kl = e / kGpRed, idx = e % kGpRed;
        int a, bb;  // idx -> (a <= bb) of the 10 x 10 upper triangle, or (a, rhs)
        if (idx >= 55) {
            a = idx - 55;
            bb = 10;
        } else {
            a = 0;
            int rem = idx;
            while (rem >= 10 - a) {
                rem -= 10 - a;
                ++a;
            }
            bb = a + rem;
        }
        const int g0 = bv.kf_gp0[wd.kf0 + kl], g1 = g0 + bv.kf_ngp[wd.kf0 + kl];  // rows are sorted by keyframe
        double acc = 0.0;
        for (int g = g0; g < g1; ++g)
            if (bv.gp_owner[g] == shard) acc += bv.gp_F[a * bv.SG + g] * (bb < 10 ? bv.gp_F[bb * bv.SG + g] : bv.gp_r[g]);
        bv.x_gp[(int64_t)(wd.kf0 + kl) * kGpRed + idx] = acc;
    }
    if (tid == 0) {
        double gmax = 0.0, xn2 = 0.0, lf = 0.0, df = 0.0, gc = 0.0;
        for (int b = wd.lblk0; b < wd.lblk0 + wd.n_lblk; ++b) {
            if (bv.lblk_owner[b] != shard) continue;
            gmax = fmax(gmax, bv.lblk_part[(int64_t)b * 8 + 0]);
            xn2 += bv.lblk_part[(int64_t)b * 8 + 1];
            if (bv.lblk_part[(int64_t)b * 8 + 5] != 0.0) df = 1.0;
            if (bv.lblk_linfail[b] != 0.0) lf = 1.0;
        }
        for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g)
            if (bv.gp_owner[g] == shard) gc += bv.gp_cost[g];
        bv.x_lb[(int64_t)w * 8 + 0] = gmax;
        bv.x_lb[(int64_t)w * 8 + 1] = xn2;
        bv.x_lb[(int64_t)w * 8 + 5] = df;
        bv.x_lf[w] = lf;
        bv.x_gc[w] = gc;
    }
}
// One double of a shard's block -> its place among the P contributions of the consumer view (after the all-gather of an
// exchange).  e = index inside the block; `cw` = the ORIGINAL window descriptors (offsets xlv_off / sred_off).
// `range` = the doubles of the block from offset `base` on (what an exchange of a sub-range of the block moved).
KBA_HD void unpack_entry(const ExchangeLayout& L, const WinDesc* cw, const double* range, size_t base, double* arena, int shard, size_t e) {
    const double v = range[e - base];
    const size_t P = (size_t)L.P, s = (size_t)shard;
    if (e < L.b_lf) {  // x_lv: window w holds [xlv_off, xlv_off + n_view * 28)
        const size_t k = e - L.b_lv;
        int w = 0;
        while (w + 1 < L.n_win && (size_t)cw[w + 1].xlv_off <= k) ++w;
        const size_t n = (size_t)cw[w].n_view * kLinPartial, in = k - (size_t)cw[w].xlv_off;
        if (in < n) arena[L.c_lv + P * (size_t)cw[w].xlv_off + s * n + in] = v;
    } else if (e < L.b_gp) {
        const size_t w = e - L.b_lf;
        if (w < (size_t)L.n_win) arena[L.c_lf + w * P + s] = v;
    } else if (e < L.b_gc) {
        const size_t k = e - L.b_gp;
        if (k < (size_t)L.TK * kGpRed) arena[L.c_gp + s * (size_t)L.TK * kGpRed + k] = v;
    } else if (e < L.b_gcc) {
        const size_t w = e - L.b_gc;
        if (w < (size_t)L.n_win) arena[L.c_gc + w * P + s] = v;
    } else if (e < L.b_lb) {
        const size_t w = e - L.b_gcc;
        if (w < (size_t)L.n_win) arena[L.c_gcc + w * P + s] = v;
    } else if (e < L.b_S) {
        const size_t k = e - L.b_lb, w = k / 8;
        if (w < (size_t)L.n_win) arena[L.c_lb + (w * P + s) * 8 + k % 8] = v;
    } else {
        const size_t k = e - L.b_S;
        int w = 0;
        while (w + 1 < L.n_win && (size_t)cw[w + 1].sred_off / P <= k) ++w;
        const size_t n = (size_t)schur_need_pad(cw[w].nf), in = k - (size_t)cw[w].sred_off / P;
        if (in < n) arena[L.c_S + (size_t)cw[w].sred_off + s * n + in] = v;
    }
}
//   after a rejected step's damping (lm_damp_lane sets the failure flag again)
KBA_HD void shard_reduce_damp(const BatchView& bv, int w, int shard) {
    const WinDesc& wd = bv.win[w];
    double df = 0.0;
    for (int b = wd.lblk0; b < wd.lblk0 + wd.n_lblk; ++b)
        if (bv.lblk_owner[b] == shard && bv.lblk_part[(int64_t)b * 8 + 5] != 0.0) df = 1.0;
    bv.x_lb[(int64_t)w * 8 + 5] = df;
}
//   after the back-substitution (before the step decision):  x_lb[2, 3, 4, 6, 7], x_gcc
KBA_HD void shard_reduce_step(const BatchView& bv, int w, int shard) {
    const WinDesc& wd = bv.win[w];
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = wd.lblk0; b < wd.lblk0 + wd.n_lblk; ++b) {
        if (bv.lblk_owner[b] != shard) continue;
        v[2] += bv.lblk_part[(int64_t)b * 8 + 2];
        v[3] += bv.lblk_part[(int64_t)b * 8 + 3];
        v[4] += bv.lblk_part[(int64_t)b * 8 + 4];
        v[6] += bv.lblk_part[(int64_t)b * 8 + 6];
        if (bv.lblk_part[(int64_t)b * 8 + 7] != 0.0) v[7] = 1.0;
    }
    double gcc = 0.0;
    for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g)
        if (bv.gp_owner[g] == shard) gcc += bv.gp_cost_c[g];
    bv.x_lb[(int64_t)w * 8 + 2] = v[2];
    bv.x_lb[(int64_t)w * 8 + 3] = v[3];
    bv.x_lb[(int64_t)w * 8 + 4] = v[4];
    bv.x_lb[(int64_t)w * 8 + 6] = v[6];
    bv.x_lb[(int64_t)w * 8 + 7] = v[7];
    bv.x_gcc[w] = gcc;
}


// ======================================================================================= trimming
// Per landmark: max over its observations of the un-robustified block norms (getMaximumResidual,
// robust_solving.cpp:82-91); < 0 when the landmark has no block in that list.
KBA_HD void trim_max_lane(const BatchView& bv, int gl, const double* plane_rep, const double* plane_dep) {
    double mr = -1.0, md = -1.0;
    if (bv.lm_state[gl]) {
        const WinDesc& wd = bv.win[bv.lm_win[gl]];
        for (int j = 0; j < wd.n_view; ++j) {
            const int s = bv.lm_slot[(int64_t)j * bv.SL + gl];
            if (s < 0) continue;
            mr = fmax(mr, plane_rep[s]);
            md = fmax(md, plane_dep[s]);
        }
    }
    bv.trim_rep[gl] = mr;
    bv.trim_dep[gl] = md;
}

// Rank-based quantile selection (TrimmerQuantile::getOutliers, trimmer_quantile.hpp:40-63): an element is an
// outlier iff its rank in (value, id) order is >= int(n_groups * quantile).  Returns 1 if landmark li (packed local
// index) is an outlier of the list `vals`; ids = the landmarks' indices in the caller's window (tie order).
KBA_HD int trim_is_outlier(const double* vals, const int32_t* ids, int n_lm, int li, double quantile, int min_groups) {
    const double v = vals[li];
    if (v < 0.0) return 0;
    int n_groups = 0, rank = 0;
    for (int j = 0; j < n_lm; ++j) {
        const double u = vals[j];
        if (u < 0.0) continue;
        ++n_groups;
        if (u < v || (u == v && ids[j] < ids[li])) ++rank;
    }
    if (n_groups < min_groups) return 0;
    const int num = (int)((double)n_groups * quantile);
    return rank >= num;
}

}  // namespace kba
