// kba_layout.hpp — HBM data layout of a batch of optimisation windows and the per-window LM state.
//
// Design: many independent windows are packed into one set of struct-of-array buffers so that
// every kernel is a flat, coalesced scan over "all observations of the batch" / "all landmarks of the batch".
//   * observations are stored VIEW-major (view = (keyframe, camera) pair): a 256-lane workgroup only ever
//     holds observations of one view, so pose and camera are wave-uniform (scalar loads) and the camera-side
//     normal-equation blocks reduce inside the workgroup;
//   * every landmark owns an ELL row  slot[j][lm] -> observation index of its measurement in view j (or -1),
//     so the landmark-parallel kernels read the same planes coalesced (neighbouring landmarks sit next to
//     each other inside every view segment);
//   * residuals/Jacobians are materialised as planes  plane[c][obs]  (8-byte coalesced stores/loads).
// The reduced camera system of a window has 10 slots per keyframe: [rot 3 | trans 3 | plane normal 3 | plane dist 1].
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/limo_hip.h"

namespace kba {

constexpr int kMaxKf = 20;        // max keyframes per window: the reference's default max_size_optimization_window
                                  // (bundle_adjuster_keyframes.hpp:129); the KITTI launch runs 12
constexpr int kMaxViews = 64;     // max (keyframe, camera) views per window (LDS tables of the Schur kernels)
constexpr int kViewLin = 64;      // doubles per view in BatchView::view_lin (kba_items.hpp:view_consts_item: 55 used)
constexpr int kCamSlots = 10;     // tangent dims per keyframe in the reduced camera system
constexpr int kMaxNc = kMaxKf * kCamSlots;
constexpr int kBlock = 256;       // lanes per workgroup in the scan kernels
constexpr int kObsPerLane = 4;    // observations per lane of the view-major kernels: the camera-side partials are
                                  // accumulated in registers over them before the one workgroup reduction
constexpr int kObsBlock = kBlock * kObsPerLane;  // observations per linearize / cost workgroup
constexpr int kSchurLm = 16;      // landmarks per Schur LDS tile (48 rows = 12 MFMA k-steps)
constexpr int kSchurLmPerBlock = 64;   // landmarks per Schur block; a wave takes SolveConsts::schur_span (1, 2 or 4)
                                       // consecutive blocks of a window and writes one partial slab
constexpr int kMaxRegRows = 1 + (kMaxKf - 1) * 5 + 3 * kMaxKf;  // scale + per pair (3+1+1) + global normal 3/kf

// number of doubles in a block partial of the linearize kernel: cost, 21 (U upper) + 6 (g)
constexpr int kGpRed = 65;  // per keyframe: upper triangle of F^T F (10 x 10: 55) + F^T r (10) of ground-plane rows (BatchView::x_gp)
constexpr int kLinPartial = 28;
constexpr int kLinWaves = 4;      // waves of a landmark workgroup: each keeps its own camera-side partial sums (no barrier per view)

// One wave of k_evaluate (evaluate-only batches): the ALIGNED range of 64 observations [base, base + 64) of the observation block
// [o0, o1) (one view; a view's observations start on a multiple of 64 in these batches); dep0 = rank, among the batch's depth
// observations in packed order, of the chunk's first depth observation.
struct EvalChunk {
    int32_t base, view, o0, o1, dep0, pad;
};

struct WinDesc {
    int32_t kf0, n_kf;
    int32_t lm0, n_lm;
    int32_t view0, n_view;
    int32_t obs0, n_obs;
    int32_t blk0, n_blk;      // linearize / cost workgroups (one view each)
    int32_t lblk0, n_lblk;    // landmark workgroups
    int32_t gp0, n_gp;
    int32_t sblk0, n_sblk;    // Schur blocks (<= kSchurLmPerBlock landmarks each)
    int32_t n_sblk_plain;     // the first n_sblk_plain of them hold landmarks WITHOUT a ground-plane row (no block straddles
                              // lm_gp0): their Schur tiles only touch the pose columns [0, nfq]
    // "fast" Schur variant: at most four keyframes with free slots and one view per keyframe (decided at pack time, per
    // window, so a window takes the same kernel alone and inside any batch)
    int32_t schur_fast, n_fk;
    int32_t fk[4];            // local keyframe index of the free keyframes
    int32_t fk_view[4];       // their view (GLOBAL view index) or -1
    int32_t nc, nc_pad;       // 10*n_kf, rounded up to 16
    int32_t nf, nf_pad;       // free camera slots (compact Schur system); nf + 1 (rhs column) rounded up to 16
    int32_t nfq;              // free POSE slots: compact indices [0,nfq) are pose slots, [nfq,nf) plane slots.  In the
                              // Schur tile / slab column nfq holds the rhs, so compact index i sits in column i + (i >= nfq)
    int32_t lm_gp0;           // first (global) landmark of the window that carries a ground-plane row (they are packed last)
    int32_t cam0;             // first global camera-slot index = kf0*10
    int32_t reg0;             // first row in the regulariser row buffers
    int32_t has_scale_reg, has_gp_reg;
    int32_t n_depth, n_repr;  // residual blocks built
    int32_t do_trim;          // n_lm > min_landmarks_for_trimming
    int32_t pose_only;        // adjustPoseOnly problem (landmarks constant)
    int32_t n_view_fixed0;    // the window's first n_view_fixed0 views belong to keyframes WITHOUT a free pose block (the Pose-fixed
                              // oldest keyframe of a sliding window): k_lin_lm forms no camera-side sums for them, k_backsub no
                              // camera-step term - their own short loops in front of the general ones (view order unchanged)
    int32_t pad_view;
    double scale_w, scale_s0; // PoseRegularization weight and target
    double speed_w, speed_dt, speed_vel[3], speed_Rb[9], speed_tb[3];  // SpeedRegularizationVector2 (pose-only)
    int64_t hcc_off;          // offset (doubles) of this window's nc x nc matrix in the Hcc buffer
    int64_t spart_off;        // offset of this window's Schur partial slabs
    int64_t sred_off;         // offset of this window's per-shard contributions in the consumer's S_red: n_shards x schur_need_pad(nf)
                              // (a shard's own block holds one of them at sred_off / n_shards)
    int64_t xlv_off;          // offset of this window's [view][kLinPartial] sums inside a shard's x_lv (the consumer's lv_part holds P
                              // of them at P * xlv_off)
    int64_t lvpart_off;       // offset of this window's camera-side partial sums in BatchView::lv_part:
                              // [landmark workgroup of the window][view][kLinPartial]
    int64_t cam_scr_off;      // >= 0: the window's camera system does not fit into LDS (more than ~12 keyframes): offset of its
                              // scratch in BatchView::cam_scratch (k_cam_assemble / k_cam_solve work there, in L2, instead)
};

// Per-window Levenberg-Marquardt state (device resident; see kba_lm.hpp).
struct WinState {
    int32_t active, need_lin, first, accept;
    int32_t iter, max_iter, term, invalid_run;
    int32_t in_phase;          // selected for the current ceres-style solve
    int32_t compute_scale;     // next linearisation defines the Jacobi scaling
    int32_t n_success, n_unsuccess;
    int32_t acc_solves, acc_iters, acc_success, last_iters;
    int32_t n_trimmed, acc_lin;
    int32_t phase, trim_round;  // streaming solve (kba_lm.hpp:sched_advance): where the window is in the solveTrimmed schedule
    int32_t redamp, pad_i;      // the last step was rejected: the landmark blocks must be damped again with the new radius
                                // (after a linearisation the landmark pass has already done it)
    double radius, decrease_factor;
    double x_cost, x_norm, fixed_cost;
    double solve_initial_cost, solve_final_cost;
    double first_initial_cost;
    double rho_pending, xnorm_pending, cost_pending;
    double gmax;
};

// Reduction inputs the LM decisions read (one entry per window).
struct WinRed {
    double lin_cost;    // cost at the linearisation point (free blocks)
    double gmax;        // |x - Plus(x, -g)|_inf
    double xnorm2;      // |x|^2 over the reduced program
    double mcc;         // model cost change of the proposed step
    double step2;       // |x - x_candidate|^2
    double cand2;       // |x_candidate|^2
    double cand_cost;   // cost at the candidate (free blocks)
    int32_t lin_fail, chol_fail, cand_fail, pad;
};

// Phases of one window in the solveTrimmed schedule (robust_solving.cpp:160-248) when every window advances on its own
// (streaming solve): trimming solve -> [retry with 3x the iterations if the cost did not decrease] -> trim -> ... ->
// final solve.
enum SchedPhase { PH_IDLE = 0, PH_TRIM_SOLVE = 1, PH_RETRY = 2, PH_TRIM = 3, PH_FINAL = 4, PH_DONE = 5 };

// Worklists the device-side scheduler rebuilds every round (streaming solve).  List k lives at
// sched_lists + sched_off[k]: element [0] = number of entries, entries from [1].
enum SchedList { SL_LBLK = 0, SL_SPLAIN, SL_SFGP, SL_SGEN, SL_WIN, SL_TBLK, SL_TLBLK, SL_TWIN, SL_COUNT };

struct SolveConsts {  // subset of limo_ba_options the kernels need
    double a_rep, a_dep;
    double inv_a_rep2, inv_a_dep2;  // 1 / a^2 (IEEE division on the host: the bits of the device's 1.0 / (a * a))
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_lm_diagonal, max_lm_diagonal, min_relative_decrease;
    int32_t max_invalid, jacobi_scaling;
    double depth_quantile, reprojection_quantile;
    int32_t min_groups, pad;
    int32_t schur_span;   // plain Schur blocks per wave in this iteration (kba_items.hpp:schur_slab_of)
    int32_t schur_span_gp;  // ground-plane Schur blocks per wave
    int32_t num_trim_rounds, trim_iters, max_iters;  // the schedule, for the device-side scheduler
    int32_t schur_nslab;  // > 0: k_cam_solve sums this many slabs from S_red instead of the window's partial slabs
    int32_t schur_packed; // 1 (landmark-sharded solve): a slab of S_red holds ONLY the entries the camera solve reads - upper triangle
                          // of the free slots + rhs, in the order cam_solve enumerates them (kba_items.hpp:schur_need_offset), stride
                          // schur_need_pad(nf) - what a shard puts on the wire per LM iteration (33 KB at 90 free slots, not 74 KB)
};

// Raw pointers to every buffer of a batch (device pointers in the library, host pointers in the emulator).
struct BatchView {
    int32_t n_win, TK, TL, TO, TV, TG, n_blk, n_lblk, n_sblk, Vmax;
    int64_t SO, SL, SG;  // plane strides (padded TO, TL, TG)
    const WinDesc* win;
    WinState* st;
    WinRed* red;
    // --- parameters: current, candidate, initial (for reset)
    double *pose, *pdir, *pdist, *lm;
    double *pose_c, *pdir_c, *pdist_c, *lm_c;
    // --- constant per keyframe / landmark / view / observation
    const int32_t* kf_win;      // [TK]
    const int32_t* kf_blk0;     // [TK] first linearize workgroup of this keyframe's views (contiguous)
    const int32_t* kf_nblk;     // [TK]
    const int32_t* kf_gp0;      // [TK] first ground-plane row attached to this keyframe (rows sorted by keyframe)
    const int32_t* kf_ngp;      // [TK]
    uint8_t* cmask;             // [TK*10] 1 = free tangent dim of the reduced program
    uint8_t* cpresent;          // [TK*10] 1 = parameter block is in the problem (free or constant)
    const int32_t* cslot;       // [TK*10] compact index of a free slot inside its window, -1 otherwise
    const int32_t* lm_win;      // [TL]
    const int32_t* lm_id;       // [TL] index of the landmark in the caller's window (ties in trimming resolve by id)
    const double* lm_weight;    // [TL]
    uint8_t* lm_state;          // [TL] 1 = in problem, 0 = removed by trimming / not constrained
    const int32_t* lm_gp;       // [TL] ground-plane residual index or -1
    const int32_t* lm_slot;     // [Vmax*SL] observation index per (view-in-window, landmark) or -1
    const int32_t* view_kf;     // [TV] global keyframe index
    const int32_t* view_win;    // [TV]
    const double* view_cam;     // [TV*16] f,cx,cy,pad, Rc[9], tc[3]
    double* view_lin_c;         // [TV*kViewLin] for the CANDIDATE poses (k_cam_solve): H, h0, intrinsics at the same places, and at
                                // [28..39] K = Rc dR (9), k0 = Rc delta_t (3) of the proposed camera step (back-substitution:
                                // F_pose delta_pose of an observation = c^T (K p + k0), no per-observation M(q, p))
    double* view_lin;           // [TV*kViewLin] per-view constants of the CURRENT poses (k_view_consts): H = Rc R(q) (9),
                                // h0 = Rc t + tc (3), Rc (9), q (4), f, cx, cy - wave-uniform operands of k_lin_lm
    const int32_t* blk_view;    // [n_blk]
    const int32_t* blk_obs0;    // [n_blk]
    const int32_t* blk_n;       // [n_blk]
    const int32_t* obs_lm;      // [TO] global landmark
    const float *obs_u, *obs_v, *obs_d;  // [TO]
    const int32_t* lblk_win;    // [n_lblk]
    const int32_t* lblk_lm0;
    const int32_t* lblk_n;
    const int32_t* sblk_win;    // [n_sblk]
    const int32_t* sblk_lm0;
    const int32_t* sblk_n;
    // --- ground-plane residuals
    // --- landmark-sharded solve: what a shard contributes per LM iteration, already summed over ITS workgroups /
    //     rows (kba_items.hpp:shard_reduce_*), in the shard's own contiguous block (kba_buffers.hpp:exchange_layout):
    const int32_t* lblk_owner;  // [n_lblk] shard that owns the landmark workgroup (null in unsharded batches)
    const int32_t* gp_owner;    // [TG] shard that owns the ground-plane row
    double* x_lv;               // [window][view][kLinPartial]  camera-side sums of the shard's landmark workgroups, in workgroup order
    double* x_lf;               // [window]     a functor failed
    double* x_gp;               // [TK][kGpRed] per keyframe: F^T F (upper, 55) | F^T r (10) of the shard's ground-plane rows, in row order
    double* x_gc;               // [window]     their cost at the linearisation point
    double* x_gcc;              // [window]     their cost at the candidate
    double* x_lb;               // [window][8]  the lblk_part entries folded over the shard's workgroups (max / sum / or)
    // ... and how the CONSUMER view (window-level kernels, replicated on every shard) sees the P contributions: through the
    // ordinary members lv_part / lblk_linfail / gp_cost / gp_cost_c / lblk_part with a WinDesc whose lblk0 / n_lblk / lvpart_off /
    // gp0 / n_gp describe "P workgroups, P rows" (one per shard, shard order) - the summation loops of cam_assemble and
    // reduce_step run unchanged - plus gp_red for the one place that consumed raw rows:
    const double* gp_red;       // [P][TK][kGpRed] or null
    int32_t gp_red_P, pad_x;
    const int32_t* gp_lm;       // [TG] global landmark
    const int32_t* gp_kf;       // [TG] global keyframe
    const double* gp_w;         // [TG] loss weight
    double *gp_r, *gp_F, *gp_E; // planes [1|10|3][SG]
    double* gp_cost;            // [TG] cost at the linearisation point
    double* gp_cost_c;          // [TG] cost at the candidate
    // --- materialised linearisation (planes over observations)
    // Solver batches keep the FACTORED Jacobian: every row of an observation is  c_row^T Rc [ M(q,p) | I ]  towards its
    // pose and  c_row^T Rc R(q)  towards its landmark, and the three c_row are spanned by FOUR scalars
    // (au, xn, yn, sd; kba_math.hpp:ft_build).  Only those and the residual are stored - 56 B per observation instead
    // of 240 B for J_pose 3x6 + J_point 3x3; the landmark-parallel kernels rebuild Ft = c^T Rc, F = Ft [M | I],
    // E = Ft R from the view / pose / landmark they hold anyway.  Evaluate-only batches (Problem::Evaluate)
    // materialise Jp / Jl in full.
    // Evaluate-only batches: the planes hold the rows that EXIST (SURVEY 8d's materialised unit: 160 B written per observation,
    // 80 B more per DEPTH observation): the reprojection rows u, v as planes over the observations, the depth row as COMPACT planes
    // over the depth observations only (PackedBatch::obs_rank: rank of the observation among those with d > 0, packed order;
    // stride SD):
    //     obs_r  = [2][SO] r_u, r_v             | [1][SD] r_d
    //     obs_Jp = [12][SO] rows u, v (2 x 6)   | [6][SD] row d
    //     obs_Jl = [6][SO] rows u, v (2 x 3)    | [3][SD] row d
    double *obs_r, *obs_c;            // obs_c = [2][SO] (au, sd), solver batches; obs_r exists in evaluate-only batches only (the solve keeps residuals in registers)
    double *obs_Jp, *obs_Jl;          // evaluate-only batches
    const EvalChunk* echunk;          // [n_echunk] evaluate-only batches: the wave-sized work items of k_evaluate
    int64_t SD;                       // stride of the compact depth-row planes
    int32_t n_echunk, pad_e;
    double* lv_part;            // camera-side partial sums of the landmark-major linearisation (WinDesc::lvpart_off)
    double* lblk_linfail;       // [n_lblk] 1.0: a functor failed in this landmark workgroup (a double: it travels in the exchange arena)
    // --- landmark side
    double *lm_V, *lm_g;        // planes [6|3][SL]  (unscaled E^T E, E^T r incl. ground-plane rows)
    double *lm_scale;           // [3][SL] Jacobi scaling
    double* lm_Li;              // planes [6][SL]  Bt = L^-1 S of the damped landmark block (V' + D^2) = L L^T (lm_damp_store)
    double* lblk_part;          // [n_lblk*8]: gmax, xnorm2, mcc, step2, cand2, fail, -, -
    // --- camera side
    double *Hcc, *gc;           // per window nc*nc (hcc_off) ; [TK*10]
    double *scale_c, *yc, *delta_c;  // [TK*10]
    double *S_part;             // Schur partial slabs
    double *S_red;              // landmark-sharded solve: one slab per (window, shard) = sum of the shard's partial slabs
    double* cam_scratch;        // scratch of the window-level kernels for windows too large for LDS (WinDesc::cam_scr_off)
    double* reg_cost;           // [n_win*2]: free / fixed regulariser cost at the linearisation point
    // --- trimming
    double *trim_rep, *trim_dep;   // [TL] max un-robustified residual norm per landmark, <0 = no block
    int32_t* n_active;          // [2] {windows still iterating, workgroups of k_cam_assemble done} of this iteration
    int32_t* n_active_host;     // pinned host word the last workgroup publishes the count to (not a batch buffer)
    // --- streaming solve (device-side scheduler, k_sched): not batch buffers, set by the library when it streams
    int32_t counted;            // 1: worklists are counted (entry [-1] of a list = its length; grids are capacities)
    int32_t n_slots;            // windows in flight at most
    int32_t* slot_win;          // [n_slots] window in the slot or -1
    int32_t* sched_ctl;         // [0] next pending window, [1] windows finished
    int32_t* slot_cnt;          // [n_slots][SL_COUNT + 1] list counts (then offsets) of the slot's window, [SL_COUNT] = kind
    int32_t* sched_lists;       // the worklists, back to back
    int32_t sched_off[SL_COUNT];  // offset of list k's count word inside sched_lists
    int32_t* sched_done_host;   // pinned ring (4 words): windows finished as of round r at [r & 3]
};

// Sizes and offsets of a landmark-sharded solve's exchange (kba_buffers.hpp:exchange_layout builds it and explains it).
struct ExchangeLayout {
    int P = 1, n_win = 0, TK = 0;
    size_t n_lv = 0, n_w = 0, n_gp = 0, n_lb = 0, n_S = 0;      // doubles per shard: x_lv, per-window scalars, x_gp, x_lb, S
    size_t b_lv = 0, b_lf = 0, b_gp = 0, b_gc = 0, b_gcc = 0, b_lb = 0, b_S = 0, b_total = 0;  // offsets inside a block
    size_t c_lv = 0, c_lf = 0, c_gp = 0, c_gc = 0, c_gcc = 0, c_lb = 0, c_S = 0, c_total = 0;  // offsets inside the consumer arena
    size_t trim_count = 0;               // doubles of [trim_rep | trim_dep]
    size_t spart_count = 1;              // doubles of the private S_part
    // range of exchange point 1 (A1), 2 (A2), 3 (A), 4 (B) inside a block
#if defined(__HIPCC__)
    __host__ __device__
#endif
    void range(int point, size_t& off, size_t& count) const {
        off = point == 2 ? b_S : point == 4 ? b_gcc : 0;
        const size_t end = point == 1 || point == 4 ? b_S : b_total;
        count = end - off;
    }
};

}  // namespace kba
