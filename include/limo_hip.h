/*
 * limo_hip.h — C-ABI of the MI355X-native LIMO hot path (keyframe bundle adjustment + LiDAR depth assignment).
 *
 * This header is the drop-in boundary.  Everything above it (window bookkeeping, landmark/keyframe
 * selection, ROS I/O) stays in the host language; everything below it runs as hand-written HIP kernels
 * on gfx950.  Plain pointers and sizes only; no C++ types, no torch types, nothing thrown across it.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference tree):
 *
 *   limo_ba_solve / limo_ba_batch_*     the body of BundleAdjusterKeyframes::solve() between "selection
 *                                       done" and "return report":
 *                                       keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:695-766
 *                                       (problem build :498-627, :769-818, :890-904; constness :198-219,
 *                                       :722-736; trimming schedule :740-758; solveTrimmed call :765) and
 *                                       robust_optimization/src/robust_solving.cpp:140-248 + ceres::Solve
 *                                       (Ceres 1.13, DENSE_SCHUR, Levenberg-Marquardt).
 *   limo_ba_adjust_pose_only            BundleAdjusterKeyframes::adjustPoseOnly(),
 *                                       bundle_adjuster_keyframes.cpp:820-888.
 *   limo_ba_evaluate                    ceres::Problem::Evaluate as used at
 *                                       robust_optimization/src/robust_solving.cpp:44 and
 *                                       keyframe_bundle_adjustment/src/definitions.cpp:94 (residuals, cost) plus
 *                                       the Jacobians Ceres builds inside Solve for the blocks created at
 *                                       bundle_adjuster_keyframes.cpp:584-620
 *                                       (internal/cost_functors_ceres.hpp:53-222).
 *   limo_landmark_init                  BundleAdjusterKeyframes::calculateLandmark (both overloads),
 *                                       bundle_adjuster_keyframes.cpp:332-382, internal/triangulator.hpp:51-75.
 *   limo_trim_quantile                  TrimmerQuantile::getOutliers,
 *                                       robust_optimization/include/robust_optimization/internal/trimmer_quantile.hpp:40-63.
 *   limo_depth_estimate                 the (un-vendored) mono_lidar_depth DepthEstimator as pinned by
 *                                       demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml:1-185;
 *                                       contract = FeaturePoint::d, matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26.
 *
 * Conventions (reference: internal/definitions.hpp:23,75-83, keyframe.hpp:181-182):
 *   pose   = (qw,qx,qy,qz,tx,ty,tz), x' = R(q) x + t, keyframe <- origin.  Camera extrinsic = camera <- vehicle.
 *   All parameters are double; measurements are float and are widened exactly as the reference does
 *   (bundle_adjuster_keyframes.cpp:585,606-607).  A measurement without depth carries d < 0 (the solve
 *   uses d > 0.0f, :578).
 *
 * Return codes: 0 = ok, < 0 = invalid input / runtime error (see limo_last_error), > 0 reserved.
 * Threading: one limo_ctx per (thread, GPU, stream); calls on one ctx are serialised by the caller.
 */
#ifndef LIMO_HIP_H_
#define LIMO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIMO_ABI_VERSION 5 /* 5: limo_ctx_comm_init_host; 2: limo_ba_evaluate_rows, limo_ctx_exchange_stats, limo_depth_last_ground_plane, limo_depth_set_timing, limo_depth_last_kernel_ms; 3: limo_ctx_coop_fallbacks; 4: limo_depth_estimate_begin / _end */

/* Keyframe::FixationStatus, keyframe.hpp:30 */
enum limo_fixation { LIMO_FIX_POSE = 0, LIMO_FIX_SCALE = 1, LIMO_FIX_NONE = 2 };

/* error codes */
enum limo_status {
    LIMO_OK = 0,
    LIMO_ERR_INVALID = -1,      /* null pointer, negative size, index out of range        */
    LIMO_ERR_NOT_ENOUGH_KF = -2, /* solve() on a window without active keyframes.  The reference's
                                    NotEnoughKeyframesException counts all PUSHED keyframes (< 3,
                                    bundle_adjuster_keyframes.cpp:630): the caller's check; the window of
                                    ACTIVE keyframes may hold 1 or 2 and is solved like the reference does */
    LIMO_ERR_RUNTIME = -3,      /* HIP runtime error                                       */
    LIMO_ERR_NO_DEVICE = -4     /* no gfx950 device / extension cannot run                 */
};

/* Termination of one Ceres-style solve (ceres::TerminationType restated). */
enum limo_termination { LIMO_CONVERGENCE = 0, LIMO_NO_CONVERGENCE = 1, LIMO_FAILURE = 2 };

/*
 * One optimisation window, flattened (struct of arrays).  The caller owns every buffer; the library keeps
 * no pointer after a call returns.  Arrays marked inout are optimised in place, exactly like Ceres does
 * through the raw pointers the reference hands it (bundle_adjuster_keyframes.cpp:592-593,555-556).
 *
 * Keyframes are the ACTIVE keyframes in ascending keyframe-id (timestamp) order — the order of
 * active_keyframe_ids_ (std::set) that the regularisers iterate (:775,:892).
 * Landmarks are the SELECTED landmarks that exist in landmarks_, ascending landmark id.
 * Observations: one per (keyframe, landmark, camera) measurement of a selected landmark in an active
 * keyframe (:569-576); any order.
 */
typedef struct limo_ba_window {
    int32_t n_kf;
    int32_t n_cam;
    int32_t n_lm;
    int32_t n_obs;

    double* kf_pose;            /* [n_kf*7] inout                                         */
    double* kf_plane_dir;       /* [n_kf*3] inout  Plane::direction                        */
    double* kf_plane_dist;      /* [n_kf]   inout  Plane::distance (< -10: no ground plane) */
    const int32_t* kf_fixation; /* [n_kf] enum limo_fixation                               */

    const double* cam;          /* [n_cam*10] f, cx, cy, qw,qx,qy,qz, tx,ty,tz (camera<-vehicle) */

    double* lm_pos;             /* [n_lm*3] inout                                          */
    const double* lm_weight;    /* [n_lm]  Landmark::weight                                */
    const uint8_t* lm_is_ground;/* [n_lm]  Landmark::is_ground_plane                       */

    const int32_t* obs_kf;      /* [n_obs] index into keyframes of this window             */
    const int32_t* obs_lm;      /* [n_obs] index into landmarks of this window             */
    const int32_t* obs_cam;     /* [n_obs] index into cameras of this window               */
    const float* obs_u;         /* [n_obs]                                                 */
    const float* obs_v;         /* [n_obs]                                                 */
    const float* obs_d;         /* [n_obs] depth along camera z, < 0 = none                */
} limo_ba_window;

/*
 * Options.  Defaults (limo_ba_default_options) are the reference's: OutlierRejectionOptions
 * (bundle_adjuster_keyframes.hpp:79-89), robust_optimization::getStandardSolverOptions
 * (robust_solving.hpp:93-108), bundle_adjuster_keyframes.cpp:740-764, and Ceres 1.13 Solver::Options defaults.
 */
typedef struct limo_ba_options {
    double depth_thres;            /* 0.16  Cauchy scale, depth blocks        */
    double reprojection_thres;     /* 1.6   Cauchy scale, reprojection blocks */
    double depth_quantile;         /* 0.95                                    */
    double reprojection_quantile;  /* 0.95                                    */
    int32_t num_trim_rounds;       /* 1   outlier_rejection_options_.num_iterations */
    int32_t trim_solver_iterations;/* 2   entries of number_iterations        */
    int32_t min_landmarks_for_trimming; /* 100: trimming only if n_lm > this (solve); 30 for pose-only */
    int32_t minimum_number_residual_groups; /* 30 */
    int32_t max_num_iterations;    /* 100 */
    double max_solver_time_sec;    /* wall-clock cap per ceres-style solve; <= 0 disables it (deterministic). */
    /* Ceres 1.13 trust-region defaults */
    double function_tolerance;     /* 1e-6  */
    double gradient_tolerance;     /* 1e-10 */
    double parameter_tolerance;    /* 1e-8  */
    double initial_trust_region_radius; /* 1e4  */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_lm_diagonal;        /* 1e-6  */
    double max_lm_diagonal;        /* 1e32  */
    double min_relative_decrease;  /* 1e-3  */
    int32_t max_num_consecutive_invalid_steps; /* 5 */
    int32_t jacobi_scaling;        /* 1 */
} limo_ba_options;

/* What the reference returns as a FullReport() string, as numbers. */
typedef struct limo_ba_report {
    int32_t termination;         /* enum limo_termination of the FINAL solve                    */
    int32_t num_solves;          /* ceres-style solves executed (trim rounds [+retries] + final) */
    int32_t iterations_total;    /* LM iterations over all solves (iteration 0 not counted)      */
    int32_t iterations_final;    /* LM iterations of the final solve                             */
    int32_t successful_steps;    /* accepted steps over all solves                               */
    int32_t n_depth_blocks;      /* residual blocks built (before trimming)                      */
    int32_t n_repr_blocks;
    int32_t n_gp_blocks;
    int32_t n_trimmed_landmarks; /* landmarks removed by trimming                                */
    int32_t num_linearizations;  /* residual+Jacobian evaluations (iteration 0 + accepted steps), all solves */
    double initial_cost;         /* cost of the first solve at x0 (incl. fixed cost)             */
    double final_cost;           /* cost after the final solve (incl. fixed cost)                */
    double time_sec;             /* wall time inside the library for this window / batch         */
} limo_ba_report;

typedef struct limo_ctx limo_ctx;
typedef struct limo_ba_batch limo_ba_batch;

/* --- context ------------------------------------------------------------------------------------ */
int limo_abi_version(void);
/* A context = one GPU + one stream + the scratch it reuses between calls.  Not thread-safe: one context per host
 * thread (several contexts may share a GPU).  Device blocks and pinned staging buffers released by finished calls stay with
 * the context for the next call and are freed by limo_ctx_destroy: up to 32 MB a few per power-of-two size class, larger ones
 * (the arena of a big batch) one per 64 MB class and at most 16 GB altogether (environment KBA_POOL_LARGE_MB sets that cap, 0
 * keeps none).  Idle blocks never cost an allocation: when the device is out of memory they are released, largest first, and
 * the allocation is tried again.  HOST side: a context that has created a batch of 128 windows or more keeps ONE page-locked
 * arena (sized for that batch, at most 1 GB) that the flattened arrays of its next large batches are written into and uploaded
 * from - one live batch at a time holds it, the others use the heap; environment KBA_NO_PACK_ARENA=1 switches it off. */
int limo_ctx_create(int device, limo_ctx** out);
void limo_ctx_destroy(limo_ctx* ctx);
/* Use an existing hipStream_t (e.g. the host framework's current stream); NULL = the context's own. */
int limo_ctx_set_stream(limo_ctx* ctx, void* hip_stream);
const char* limo_last_error(const limo_ctx* ctx);
/* Page-locked host memory for buffers that cross PCIe every call (a frame's sweep, a window's observations): every
 * entry point takes pageable or page-locked host pointers alike, page-locked ones are read by the GPU's copy engine
 * directly instead of through the runtime's staging copy.  NULL when the allocation fails. */
void* limo_host_alloc(size_t bytes);
void limo_host_free(void* p);

void limo_ba_default_options(limo_ba_options* out);

/* --- bundle adjustment --------------------------------------------------------------------------- */
/* One window, host buffers in, optimised in place.  = batch_create(1) + solve + download + destroy.
 * The call the reference's node makes per keyframe (mono_lidar.cpp:255).  A window of the common shape (<= 4 free
 * keyframes, one camera each) is solved in ONE cooperative kernel launch (k_solve_coop: several workgroups that meet at
 * device-wide barriers), any other window as a launch sequence - same results bit for bit (KBA_NO_COOP_SOLVE=1 forces
 * the launch sequence).  A wall-clock cap (opts->max_solver_time_sec > 0) is honoured on both paths. */
int limo_ba_solve(limo_ctx* ctx, limo_ba_window* window, const limo_ba_options* opts, limo_ba_report* report);

/*
 * Many independent windows per launch sequence (ragged sizes allowed).  create() packs and uploads;
 * solve() runs entirely from HBM-resident data; reset() restores the uploaded initial parameters on the
 * device (so a batch can be re-solved, e.g. for benchmarking); download() writes the optimised parameters
 * back into the callers' windows (same shapes as at create) and fills n reports (either may be NULL).
 */
int limo_ba_batch_create(limo_ctx* ctx, int32_t n_windows, const limo_ba_window* windows, limo_ba_batch** out);
int limo_ba_batch_solve(limo_ba_batch* batch, const limo_ba_options* opts);
int limo_ba_batch_reset(limo_ba_batch* batch);
int limo_ba_batch_download(limo_ba_batch* batch, limo_ba_window* windows_out, limo_ba_report* reports);
void limo_ba_batch_destroy(limo_ba_batch* batch);
/* Which landmarks of window `window` the trimming rounds of the last solve removed (the outlier groups
 * robust_optimization::solveTrimmed erases, robust_solving.cpp:183-215; the reference only logs their number):
 * removed[l] = 1 for landmark l (the caller's landmark order), else 0.  removed has n_lm entries. */
int limo_ba_batch_trimmed(limo_ba_batch* batch, int32_t window, uint8_t* removed);
/* Device time (ms, HIP events on the batch's stream) and launch count of the Jacobian-evaluation kernel
 * accumulated since create()/the last call with reset != 0; either output may be NULL. */
int limo_ba_batch_kernel_stats(limo_ba_batch* batch, int reset, double* linearize_ms, int64_t* linearize_launches,
                               double* total_ms);
/* Same accumulation window, per timed kernel (the two that dominate an LM iteration). */
#define LIMO_KERNEL_LINEARIZE 0 /* k_lin_lm: residuals + (factored) Jacobians of every observation, landmark blocks */
#define LIMO_KERNEL_SCHUR 1     /* k_schur_lean / k_schur_wide: Schur complement of the landmark blocks (f64 MFMA) */
int limo_ba_batch_kernel_time(limo_ba_batch* batch, int kernel, double* ms, int64_t* launches);

/*
 * Landmark-sharded solve of ONE (large) window, SURVEY §8e / BASELINE.json configs[3]: shard s owns the landmarks
 * whose index in the window satisfies  index mod n_shards == s  together with all their observations and
 * ground-plane rows; camera-side parameters are replicated.  Per LM iteration a shard contributes ONE contiguous block -
 * its camera-side sums per view, its ground-plane rows folded per keyframe (F^T F | F^T r) and the entries of [S | rhs] the
 * camera solve reads (upper triangle of the free slots + rhs): 42 KB for a 10-keyframe / 8000-landmark window - exchanged
 * by ONE all-gather before camera assembly + camera solve (two pieces of it in the first iteration of a solve, where the
 * assembly defines the Jacobi scale the Schur complement needs), and nine doubles by a second all-gather before the step
 * decision; one all-reduce per trimming round and one at the end for the landmarks.  Then every shard factors the reduced
 * camera system redundantly and back-substitutes its own landmarks.  Nothing is summed on the wire: every rank adds the P
 * contributions in shard order, so the result does not depend on how the shards are spread over ranks.
 *   - with a communicator (limo_ctx_comm_init): one process per GPU, shard s lives on rank s mod world (n_shards a
 *     multiple of world, normally == world); every rank calls with the same window; the exchange is an RCCL
 *     all-gather of the blocks (one call per local shard); all ranks return the full result;
 *   - without: n_shards (<= 8) VIRTUAL shards on this GPU - the same kernels, the same blocks, no wire
 *     (the single-GPU mode SURVEY §8e asks for to check the sharding arithmetic).
 * Replaces the same reference call as limo_ba_solve (bundle_adjuster_keyframes.cpp:629-767).
 */
#define LIMO_COMM_ID_BYTES 128
int limo_comm_unique_id(unsigned char id[LIMO_COMM_ID_BYTES]);            /* rank 0; broadcast the bytes to all ranks */
int limo_ctx_comm_init(limo_ctx* ctx, const unsigned char id[LIMO_COMM_ID_BYTES], int rank, int world);
/* The same exchange through a transport of the CALLER instead of RCCL: every exchange step is staged through page-locked host
 * memory and handed to `fn` - kind 0: all-gather (send = count doubles of this rank, recv = world * count doubles in rank order),
 * kind 1: sum over the ranks (send = recv = count doubles).  For ranks that cannot form an RCCL communicator - two processes
 * sharing ONE GPU (a device may appear once per communicator), a host-side fabric - and for tests of the multi-rank branch on a
 * one-GPU box.  fn = NULL removes the transport; limo_ctx_comm_init replaces it. */
typedef void (*limo_exchange_fn)(const double* send, double* recv, long long count, int kind, void* user);
int limo_ctx_comm_init_host(limo_ctx* ctx, limo_exchange_fn fn, void* user, int rank, int world);
int limo_ba_solve_sharded(limo_ctx* ctx, limo_ba_window* window, const limo_ba_options* opts, int n_shards,
                          limo_ba_report* report);
/* Exchange accounting of the last limo_ba_solve_sharded on this context: stats3 = (exchange steps = collective calls with a
 * communicator, counted per local shard; bytes this rank's shards put into them; LM iterations of the solve). */
int limo_ctx_exchange_stats(limo_ctx* ctx, int64_t* stats3);
/* How many one-launch solves (k_solve_coop behind limo_ba_solve / small batches) on this context gave up at a device-wide
 * barrier and were redone as a launch sequence.  A barrier waits at most KBA_COOP_TIMEOUT_MS (default 50) of the GPU's
 * constant clock, which keeps running while a wave is preempted (shared GPU, debugger, profiler); the redo starts from the
 * batch's initial state and gives the same results - the caller never sees an error, this counter is how it can tell. */
int64_t limo_ctx_coop_fallbacks(const limo_ctx* ctx);

/*
 * Evaluate the reprojection / depth residual blocks of a window at its current parameters
 * (Problem::Evaluate semantics).  Outputs are per observation in the caller's observation order; rows
 * 0,1 = reprojection (u,v), row 2 = depth (zero when the observation has no depth):
 *   residuals [n_obs*3]
 *   jac_pose  [n_obs*3*6]  d r / d (rotation tangent 3, translation 3)  — local parameterisation applied
 *   jac_lm    [n_obs*3*3]  d r / d landmark
 *   valid     [n_obs]      0 where the reference functor returns false (|z| < 0.01, cost_functors_ceres.hpp:78-83)
 * apply_loss != 0 applies the Ceres corrector (sqrt(rho') scaling) of the block's loss; cost = 1/2 sum rho.
 * Any output pointer may be NULL.
 */
int limo_ba_evaluate(limo_ctx* ctx, const limo_ba_window* window, const limo_ba_options* opts, int apply_loss,
                     double* cost, double* residuals, double* jac_pose, double* jac_lm, uint8_t* valid);

/* Measurement aid (bench.py `roofline_evaluate`): the same evaluation (apply_loss = 1) over a batch of windows kept on
 * the device, `reps` launches timed with HIP events on the context's stream; nothing is downloaded.
 * device_ms = average time of one launch over all windows. */
int limo_ba_evaluate_batch_time(limo_ctx* ctx, int32_t n_windows, const limo_ba_window* windows, const limo_ba_options* opts,
                                int32_t reps, double* device_ms);

/*
 * Motion-only adjustment of ONE new keyframe against fixed landmarks (adjustPoseOnly).
 * window: n_kf == 1 (the new keyframe; its pose is optimised in place), landmarks constant.
 * Optional speed prior (SpeedRegularizationVector2, cost_functors_ceres.hpp:300-353): enabled when
 * speed_weight > 0: residual (t_new - R_new R_b^T t_b)/dt_cur - vel_prev with pose_before = (q_b,t_b).
 * The call the reference's node makes for every frame (mono_lidar.cpp:203): the whole solveTrimmed schedule of the
 * window runs in ONE kernel launch of one workgroup (k_solve_wg; KBA_NO_WG_SOLVE=1 KBA_NO_COOP_SOLVE=1 force the
 * launch sequence - same bits).
 */
typedef struct limo_speed_prior {
    double speed_weight;     /* 1 - rot_diff/0.03, <= 0 disables (bundle_adjuster_keyframes.cpp:842-843) */
    double dt_cur;           /* seconds                                                                */
    double vel_prev[3];      /* translation(pose_before * pose_before2^-1) / dt_before                  */
    double pose_before[7];
} limo_speed_prior;
int limo_ba_adjust_pose_only(limo_ctx* ctx, limo_ba_window* window, const limo_speed_prior* prior,
                             const limo_ba_options* opts, limo_ba_report* report);

/*
 * The residual rows of a window that are NOT reprojection / depth blocks, linearised at the window's current parameters the
 * way the solve linearises them (loss corrector / sqrt(weight) applied; Jacobians towards the TANGENT of every parameter block
 * the residual block touches, constant blocks included - the solve masks those later):
 *   LIMO_ROW_GROUND_HEIGHT  GroundPlaneHeightRegularization, cost_functors_ceres.hpp:358-385, wired by
 *                           addGroundPlaneResiduals, bundle_adjuster_keyframes.cpp:517-562 (nearest keyframe, Huber(0.1)
 *                           scaled by 10 (1 - dist / 25)); one row per selected ground landmark that gets a block
 *   LIMO_ROW_SCALE          PoseRegularization, cost_functors_ceres.hpp:229-243, wiring :890-904, weight :704-716
 *   LIMO_ROW_NORMAL_DIFF    VectorDifferenceRegularization (3 rows, sub = component), :775-784
 *   LIMO_ROW_DIST_DIFF      GroundPlaneDistanceRegularization, :786-791
 *   LIMO_ROW_PLANE_MOTION   GroundPlaneMotionRegularization, cost_functors_ceres.hpp:533-547, wiring :794-800
 *   LIMO_ROW_GLOBAL_NORMAL  VectorDifferenceRegularization2 against (0,0,1) (3 rows), :809-816
 *   LIMO_ROW_SPEED          SpeedRegularizationVector2 (3 rows; adjustPoseOnly problem only), cost_functors_ceres.hpp:300-353,
 *                           wiring :835-853
 * kf[0] < kf[1] are the window's keyframe indices the block touches (kf[1] = -1: one keyframe); jac_kf[i] holds the ten
 * tangent slots of kf[i]: rotation 3 | translation 3 | plane normal 3 | plane distance 1.  fixed = 1: every parameter block
 * of the residual block is constant (its cost is part of the fixed cost).  pose_only != 0 builds the adjustPoseOnly problem
 * (window->n_kf == 1, landmarks constant, `prior` as in limo_ba_adjust_pose_only).  Rows come in the order: ground rows
 * (ascending keyframe, then as packed), scale, per consecutive keyframe pair [normal diff x3, dist diff, plane motion],
 * per keyframe global normal x3, speed x3.  *n_rows receives the number of rows of the problem; at most cap are written.
 * The evaluation runs on the device through the same device functions the solve kernels call (gp_lane, reg_row_eval).
 */
enum limo_row_kind {
    LIMO_ROW_GROUND_HEIGHT = 0,
    LIMO_ROW_SCALE = 1,
    LIMO_ROW_NORMAL_DIFF = 2,
    LIMO_ROW_DIST_DIFF = 3,
    LIMO_ROW_PLANE_MOTION = 4,
    LIMO_ROW_GLOBAL_NORMAL = 5,
    LIMO_ROW_SPEED = 6
};
typedef struct limo_ba_row {
    int32_t kind;          /* enum limo_row_kind                                                   */
    int32_t sub;           /* component of a multi-row block                                       */
    int32_t kf[2];         /* window keyframe indices (ascending), kf[1] = -1 if only one          */
    int32_t lm;            /* landmark index in the caller's window (ground rows), else -1         */
    int32_t fixed;         /* 1 = all parameter blocks of the block are constant                   */
    double r;              /* residual, corrector applied                                          */
    double cost;           /* 1/2 rho(|r_block|^2) of the BLOCK, reported on its sub == 0 row      */
    double jac_kf[2][10];  /* d r / d tangent of kf[0], kf[1]                                      */
    double jac_lm[3];      /* d r / d landmark (ground rows)                                       */
} limo_ba_row;
int limo_ba_evaluate_rows(limo_ctx* ctx, const limo_ba_window* window, const limo_speed_prior* prior, int pose_only,
                          const limo_ba_options* opts, int32_t cap, limo_ba_row* rows, int32_t* n_rows);

/* --- landmark initialisation ----------------------------------------------------------------------- */
/*
 * Runs on the device (one lane per landmark, limo_amd/csrc/landmark_init.hip): hand over ALL new landmarks of a
 * keyframe in one call.  ctx is required.
 * For each of n landmarks: rays are given as n_rays_off CSR over (pose_cam_origin [7], u, v, d, f, cx, cy).
 * If the FIRST listed measurement with d >= 0 exists in the landmark's own (newest) keyframe the caller
 * should pass it as mode 0 (depth back-projection, :332-355); else mode 1 = midpoint triangulation over all
 * rays (:358-382; needs >= 2 rays).  ok[i] = 0 where no position could be computed.
 */
typedef struct limo_ray {
    double pose_cam_origin[7]; /* camera <- origin = T_cam_veh * T_kf_origin */
    double f, cx, cy;
    float u, v, d;
    float pad;
} limo_ray;
int limo_landmark_init(limo_ctx* ctx, int32_t n, const int32_t* ray_off /* [n+1] */, const limo_ray* rays,
                       const uint8_t* use_depth /* [n] */, double* pos_out /* [n*3] */, uint8_t* ok /* [n] */);

/* --- trimming --------------------------------------------------------------------------------------- */
/* Quantile trimmer: n (id, value) pairs; outliers = everything from sorted position int(n*q) upwards
 * (ties broken by id so the result is deterministic).  Returns the number of outliers written. */
int limo_trim_quantile(int32_t n, const int64_t* ids, const double* values, double quantile, int64_t* outliers_out);

/* --- LiDAR depth assignment -------------------------------------------------------------------------- */
/* Parameters mirror mono_lidar_fusion_parameters.yaml key by key (defaults = that file). */
typedef struct limo_depth_params {
    int32_t pixelarea_search_width;        /* 6   yaml:14 */
    int32_t pixelarea_search_height;       /* 9   yaml:17 */
    int32_t pixelarea_search_offset_x;     /* 0   yaml:21 */
    int32_t pixelarea_search_offset_y;     /* 0   yaml:24 */
    int32_t neighbors_count_min;           /* 3   yaml:48 */
    int32_t do_use_histogram_segmentation; /* 1   yaml:58 */
    double histogram_segmentation_bin_width;   /* 0.3 yaml:61 */
    int32_t histogram_segmentation_min_pointcount; /* 1 yaml:63 */
    int32_t treshold_depth_enabled;        /* 1   yaml:97  */
    double treshold_depth_max;             /* 100 yaml:101 */
    double treshold_depth_min;             /* 0   yaml:103 */
    int32_t treshold_depth_local_enabled;  /* 1   yaml:108 */
    int32_t treshold_depth_local_valuetype;/* 1 = relative yaml:112 */
    double treshold_depth_local_value;     /* 0.5 yaml:114 */
    int32_t do_use_cut_behind_camera;      /* 1   yaml:168 */
    int32_t do_use_triangle_size_maximation; /* 1 yaml:171 */
    int32_t do_check_triangleplanar_condition; /* 1 yaml:173 */
    double triangleplanar_crossnorm_treshold;  /* 0.1 yaml:176 */
    double viewray_plane_orthoganality_treshold; /* 0.1 yaml:178 */
    /* ground plane (yaml:125-163) */
    int32_t do_use_ransac_plane;           /* 1   */
    double ransac_plane_distance_treshold; /* 0.2 */
    double ransac_plane_min_z;             /* -3.5 (lidar frame) */
    double ransac_plane_max_z;             /* -1.0 */
    int32_t ransac_plane_max_iterations;   /* 600 */
    double ransac_plane_probability;       /* 0.99 */
    int32_t ransac_plane_use_refinement;   /* 1   */
    double ransac_plane_refinement_treshold;   /* 10.2 */
    double ransac_plane_point_distance_treshold; /* 0.2 */
    int32_t plane_estimator_use_mestimator; /* 1  */
    uint64_t ransac_seed;                  /* ours: RANSAC sampling seed (deterministic) */
} limo_depth_params;
void limo_depth_default_params(limo_depth_params* out);

/*
 * Assign a depth to each feature of one frame from one LiDAR sweep.
 *   cloud_xyzi  [n_pts*4] float, KITTI velodyne .bin layout (x,y,z,intensity), lidar frame
 *   T_cam_lidar [7]       camera <- lidar
 *   feat_uv     [n_feat*2] float pixel coordinates
 *   feat_is_ground [n_feat] optional (NULL = none): features labelled as ground use the ground-plane path
 *   depth_out   [n_feat]  metres along camera z, -1 where no depth could be assigned
 */
int limo_depth_estimate(limo_ctx* ctx, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar,
                        double f, double cx, double cy, int32_t img_w, int32_t img_h, const float* feat_uv,
                        size_t n_feat, const uint8_t* feat_is_ground, const limo_depth_params* params,
                        float* depth_out);

/*
 * The same call in two halves, so that a frame's depth assignment runs on the GPU while the host thread is busy with something
 * else (the reference runs the depth estimator as a process of its own beside the bundle-adjustment node:
 * demo_keyframe_bundle_adjustment_meta/launch/kitti_standalone.launch:14-23 the `tracklet_depth_node`, :55 the BA node - the two overlap there too):
 *   limo_depth_estimate_begin  enqueues the copies and the kernels on the context's stream and returns; cloud_xyzi must stay
 *                              valid until _end (from limo_host_alloc memory the copy is a DMA, from pageable memory the runtime
 *                              stages it before returning); feat_uv / feat_is_ground are consumed before it returns
 *   limo_depth_estimate_end    waits for that work and writes depth_out[n_feat] (n_feat of the _begin call)
 * One call may be open per context: _begin with one open, or _end (or any other depth call) out of order -> LIMO_ERR_INVALID.
 * limo_depth_estimate is exactly _begin followed by _end.
 */
int limo_depth_estimate_begin(limo_ctx* ctx, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar,
                              double f, double cx, double cy, int32_t img_w, int32_t img_h, const float* feat_uv,
                              size_t n_feat, const uint8_t* feat_is_ground, const limo_depth_params* params);
int limo_depth_estimate_end(limo_ctx* ctx, float* depth_out, size_t n_feat);

/*
 * The same for a batch of sweeps of one sensor rig (an offline replay: the reference's demo application reads every
 * frame of a KITTI sequence from disk, demo_keyframe_bundle_adjustment_meta/apps/main_program/main_program.cpp:97-170):
 * the frame is a grid dimension of every kernel, so a call costs seven launches whatever n_frames is.  Results are
 * those of n_frames separate limo_depth_estimate calls, bit for bit.
 *   flags & LIMO_DEPTH_DEVICE_POINTERS: cloud_xyzi / feat_uv / feat_is_ground / depth_out of every frame are device
 *   pointers on the context's GPU (sweeps already resident in HBM; nothing is copied).  A non-NULL feat_is_ground then
 *   means "estimate the ground plane of this frame" (the labels cannot be inspected from the host).
 */
typedef struct limo_depth_frame {
    const float* cloud_xyzi;       /* [n_pts*4]  */
    size_t n_pts;
    const float* feat_uv;          /* [n_feat*2] */
    size_t n_feat;
    const uint8_t* feat_is_ground; /* [n_feat] or NULL */
    float* depth_out;              /* [n_feat]   */
} limo_depth_frame;
#define LIMO_DEPTH_DEVICE_POINTERS 1u
int limo_depth_estimate_batch(limo_ctx* ctx, int32_t n_frames, const limo_depth_frame* frames, const double* T_cam_lidar,
                              double f, double cx, double cy, int32_t img_w, int32_t img_h, const limo_depth_params* params,
                              uint32_t flags);

/*
 * Ground plane of frame `frame` of the LAST limo_depth_estimate / limo_depth_estimate_batch launch group on this context
 * (a batch call runs groups of 32 frames; `frame` counts inside the last group): plane4 = (n, d) with n.p + d = 0 in the
 * camera frame, d >= 0; all zero and *inliers = 0 if the frame had no ground-labelled feature or no plane was found.
 * The estimator computes this plane anyway (yaml:125-143); the accessor exists so that a caller - and the parity tests -
 * can look at it.
 */
int limo_depth_last_ground_plane(limo_ctx* ctx, int32_t frame, double* plane4, int32_t* inliers /* may be NULL */);
/*
 * Measurement aid (bench.py `roofline_depth`): with timing on, every launch group records HIP events on the context's
 * stream around its kernels; limo_depth_last_kernel_ms returns the device time of the last group in milliseconds:
 * ms4 = (k_project, ground-plane kernels k_ransac / k_pick / k_refine / k_plane, k_features, their sum).
 */
int limo_depth_set_timing(limo_ctx* ctx, int32_t on);
int limo_depth_last_kernel_ms(limo_ctx* ctx, double* ms4);

#ifdef __cplusplus
}
#endif
#endif /* LIMO_HIP_H_ */
