// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// kba_oracle.cpp — CPU restatement of the keyframe-BA hot path on top of ceres_like:
//   build_solve_problem   BundleAdjusterKeyframes::solve() problem construction,
//                         keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:695-736
//                         (addActiveKeyframesToProblem :498-515, addKeyframeToProblem :564-627,
//                          addGroundPlaneResiduals :517-562, addScaleRegularization :890-904,
//                          addGroundplaneRegularization :769-818, deactivatePoseParameters :198-219)
//   solve_trimmed         robust_optimization/src/robust_solving.cpp:140-248 (+ :16-137)
//   trim_quantile         robust_optimization/include/robust_optimization/internal/trimmer_quantile.hpp:40-63
//   pose-only             bundle_adjuster_keyframes.cpp:820-888
//   landmark init         bundle_adjuster_keyframes.cpp:332-382, internal/triangulator.hpp:51-75
// It consumes the same flat limo_ba_window the HIP library consumes (include/limo_hip.h) so both run on
// identical bytes.  Exposed with C linkage for ctypes (tests, bench cpu_baseline).
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <set>

#include "../include/limo_hip.h"
#include "ceres_like.hpp"

using namespace kba_oracle;

namespace {

using IdMap = std::map<ResidualBlock*, std::pair<unsigned long, int>>;  // ResidualIdMap, robust_solving.hpp:26

struct RowMeta {  // a residual block that is not a reprojection / depth block (oracle_ba_evaluate_rows)
    ResidualBlock* rb;
    int kind;  // enum limo_row_kind
    int lm;    // landmark of a ground-height block, else -1
};
struct Built {
    Problem problem;
    IdMap depth, repr, gp;
    int n_depth = 0, n_repr = 0, n_gp = 0;
    std::vector<RowMeta> other;
};

Pose7 cam_pose(const double* cam10) {
    Pose7 p;
    for (int i = 0; i < 7; ++i) p[i] = cam10[3 + i];
    return p;
}

// addKeyframeToProblem, :564-627 for the observations of a window (any order).
void add_observations(const limo_ba_window& w, const limo_ba_options& o, Built& B, bool landmarks_constant) {
    for (int i = 0; i < w.n_obs; ++i) {
        const int k = w.obs_kf[i], l = w.obs_lm[i], c = w.obs_cam[i];
        const double* cam = w.cam + 10 * c;
        ParamBlock* pose = B.problem.AddParameterBlock(w.kf_pose + 7 * k, 7, PK_POSE_QUAT_R3);
        ParamBlock* lm = B.problem.AddParameterBlock(w.lm_pos + 3 * l, 3, PK_EUCLIDEAN, !landmarks_constant);
        if (w.obs_d[i] > 0.0f) {  // :578
            LandmarkDepthError f{static_cast<double>(w.obs_d[i]), cam_pose(cam)};
            auto* cost = new AutoDiffCost<LandmarkDepthError, 1, 7, 3>(f);
            ResidualBlock* rb =
                B.problem.AddResidualBlock(cost, Loss::ScaledCauchy(o.depth_thres, w.lm_weight[l]), {pose, lm});
            B.depth[rb] = std::make_pair((unsigned long)l, 1);
        }
        ReprojectionErrorWithQuaternions f{static_cast<double>(w.obs_u[i]), static_cast<double>(w.obs_v[i]), cam[0],
                                           cam[1], cam[2], cam_pose(cam)};
        auto* cost = new AutoDiffCost<ReprojectionErrorWithQuaternions, 2, 7, 3>(f);
        ResidualBlock* rb =
            B.problem.AddResidualBlock(cost, Loss::ScaledCauchy(o.reprojection_thres, w.lm_weight[l]), {pose, lm});
        B.repr[rb] = std::make_pair((unsigned long)l, 2);
    }
    B.n_depth = (int)B.depth.size();
    B.n_repr = (int)B.repr.size();
}

double transl_norm_T1_T0inv(const double* p1, const double* p0) {
    Iso<double> a = convert(p1), b = convert(p0);
    Iso<double> d = compose(a, inverse(b));
    return std::sqrt(d.t[0] * d.t[0] + d.t[1] * d.t[1] + d.t[2] * d.t[2]);
}

void build_solve_problem(const limo_ba_window& w, const limo_ba_options& o, Built& B) {
    add_observations(w, o, B, false);
    Problem& P = B.problem;
    // addGroundPlaneResiduals(10.), :517-562
    const double weight = 10.;
    for (int l = 0; l < w.n_lm; ++l) {
        if (!w.lm_is_ground[l]) continue;
        double min_dist = std::numeric_limits<double>::max();
        int kf_id = -1;
        for (int k = 0; k < w.n_kf; ++k) {
            if (w.kf_plane_dist[k] < -10.) continue;
            Iso<double> T = convert(w.kf_pose + 7 * k);
            double q[3];
            apply(T, w.lm_pos + 3 * l, q);
            double dist = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            if (dist < min_dist) {
                min_dist = dist;
                kf_id = k;
            }
        }
        if (min_dist == std::numeric_limits<double>::max()) continue;
        const double max_valid_dist = 25.;
        if (min_dist < max_valid_dist) {
            const double robust_loss_scale = 0.1;
            const double loss_weight = weight * (1. - min_dist / max_valid_dist);
            auto* cost = new AutoDiffCost<GroundPlaneHeightRegularization, 1, 7, 3, 1, 3>(GroundPlaneHeightRegularization());
            ParamBlock* pose = P.AddParameterBlock(w.kf_pose + 7 * kf_id, 7, PK_POSE_QUAT_R3);
            ParamBlock* dir = P.AddParameterBlock(w.kf_plane_dir + 3 * kf_id, 3, PK_FIX_SCALE_VECTOR);
            ParamBlock* dist = P.AddParameterBlock(w.kf_plane_dist + kf_id, 1, PK_EUCLIDEAN);
            ParamBlock* lm = P.AddParameterBlock(w.lm_pos + 3 * l, 3, PK_EUCLIDEAN, true);
            ResidualBlock* rb =
                P.AddResidualBlock(cost, Loss::ScaledHuber(robust_loss_scale, loss_weight), {pose, dir, dist, lm});
            B.gp[rb] = std::make_pair((unsigned long)l, 1);
            B.other.push_back({rb, LIMO_ROW_GROUND_HEIGHT, l});
        }
    }
    B.n_gp = (int)B.gp.size();

    auto add_scale_reg = [&](double wgt) {  // addScaleRegularization, :890-904
        if (w.n_kf > 1) {
            double current_scale = transl_norm_T1_T0inv(w.kf_pose + 7, w.kf_pose);
            auto* cost = new AutoDiffCost<PoseRegularization, 1, 7, 7>(PoseRegularization{current_scale});
            ParamBlock* p1 = P.AddParameterBlock(w.kf_pose + 7, 7, PK_POSE_QUAT_R3);
            ParamBlock* p0 = P.AddParameterBlock(w.kf_pose, 7, PK_POSE_QUAT_R3);
            B.other.push_back({P.AddResidualBlock(cost, Loss::ScaledTrivial(wgt), {p1, p0}), LIMO_ROW_SCALE, -1});
        }
    };
    // :704-716
    if (B.n_depth > 10 || B.n_gp > 10) {
        if (B.n_gp < 30) {
            double scale_reg_weight = 1000. / (static_cast<double>(B.n_depth + static_cast<double>(B.n_gp)));
            add_scale_reg(scale_reg_weight);
        }
    } else {
        add_scale_reg(1000.);
    }
    // :717-719 addGroundplaneRegularization(10.), :769-818
    if (B.n_gp > 0 && w.n_kf > 1) {
        const double wgt = 10.;
        for (int k0 = 0; k0 + 1 < w.n_kf; ++k0) {
            const int k1 = k0 + 1;
            ParamBlock* d1 = P.AddParameterBlock(w.kf_plane_dir + 3 * k1, 3, PK_FIX_SCALE_VECTOR);
            ParamBlock* d0 = P.AddParameterBlock(w.kf_plane_dir + 3 * k0, 3, PK_FIX_SCALE_VECTOR);
            B.other.push_back({P.AddResidualBlock(new AutoDiffCost<VectorDifferenceRegularization, 3, 3, 3>(VectorDifferenceRegularization()),
                                                  Loss::ScaledTrivial(3. * wgt), {d1, d0}),
                               LIMO_ROW_NORMAL_DIFF, -1});
            ParamBlock* h1 = P.AddParameterBlock(w.kf_plane_dist + k1, 1, PK_EUCLIDEAN);
            ParamBlock* h0 = P.AddParameterBlock(w.kf_plane_dist + k0, 1, PK_EUCLIDEAN);
            B.other.push_back({P.AddResidualBlock(
                                   new AutoDiffCost<GroundPlaneDistanceRegularization, 1, 1, 1>(GroundPlaneDistanceRegularization()),
                                   Loss::ScaledTrivial(wgt), {h1, h0}),
                               LIMO_ROW_DIST_DIFF, -1});
            ParamBlock* p0 = P.AddParameterBlock(w.kf_pose + 7 * k0, 7, PK_POSE_QUAT_R3);
            ParamBlock* p1 = P.AddParameterBlock(w.kf_pose + 7 * k1, 7, PK_POSE_QUAT_R3);
            B.other.push_back({P.AddResidualBlock(
                                   new AutoDiffCost<GroundPlaneMotionRegularization, 1, 7, 7, 3>(GroundPlaneMotionRegularization()),
                                   Loss::ScaledTrivial(2. * wgt), {p0, p1, d0}),
                               LIMO_ROW_PLANE_MOTION, -1});
        }
        for (int k = 0; k < w.n_kf; ++k) {
            ParamBlock* d = P.AddParameterBlock(w.kf_plane_dir + 3 * k, 3, PK_FIX_SCALE_VECTOR);
            B.other.push_back({P.AddResidualBlock(new AutoDiffCost<VectorDifferenceRegularization2, 3, 3>(
                                                      VectorDifferenceRegularization2{{0., 0., 1.}}),
                                                  Loss::ScaledTrivial(wgt), {d}),
                               LIMO_ROW_GLOBAL_NORMAL, -1});
        }
    }
    // :722-728 fix plane distance if it is the only scale information
    if (B.n_depth < 10) {
        for (int k = 0; k < w.n_kf; ++k)
            if (ParamBlock* h = P.Get(w.kf_plane_dist + k)) h->constant = true;
    }
    // :736 deactivatePoseParameters({Pose}), :198-219
    for (int k = 0; k < w.n_kf; ++k) {
        if (w.kf_fixation[k] != LIMO_FIX_POSE) continue;
        if (ParamBlock* p = P.Get(w.kf_pose + 7 * k)) p->constant = true;
        if (ParamBlock* p = P.Get(w.kf_plane_dir + 3 * k)) p->constant = true;
        if (ParamBlock* p = P.Get(w.kf_plane_dist + k)) p->constant = true;
    }
}

// adjustPoseOnly problem, :820-862: the observations of ONE keyframe against constant landmarks + the speed prior
void build_pose_only_problem(const limo_ba_window& w, const limo_speed_prior* prior, const limo_ba_options& o, Built& B) {
    add_observations(w, o, B, true);
    for (auto& p : B.problem.params)
        if (p->size == 3) p->constant = true;  // deactivateLandmarks(), :862
    if (prior && prior->speed_weight > 0.0) {  // :835-853
        SpeedRegularizationVector2 f;
        f.dt_cur_ = prior->dt_cur;
        for (int i = 0; i < 3; ++i) f.vel_before_before2_[i] = prior->vel_prev[i];
        f.pose_origin_before_eigen_ = inverse(convert(prior->pose_before));
        ParamBlock* pose = B.problem.AddParameterBlock(w.kf_pose, 7, PK_POSE_QUAT_R3);
        B.other.push_back({B.problem.AddResidualBlock(new AutoDiffCost<SpeedRegularizationVector2, 3, 7>(f),
                                                      Loss::ScaledTrivial(prior->speed_weight), {pose}),
                           LIMO_ROW_SPEED, -1});
    }
}

// TrimmerQuantile::getOutliers.  std::nth_element leaves ties unspecified; we order by (value, id).
std::vector<unsigned long> trim_quantile(const std::map<unsigned long, double>& in, double q) {
    std::vector<std::pair<unsigned long, double>> v(in.begin(), in.end());
    int num = static_cast<int>(static_cast<double>(v.size()) * q);
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) {
        return a.second < b.second || (a.second == b.second && a.first < b.first);
    });
    std::vector<unsigned long> out;
    for (size_t i = (size_t)std::max(num, 0); i < v.size(); ++i) out.push_back(v[i].first);
    return out;
}

// getResidualsToRemove, robust_solving.cpp:100-125 (calculateResiduals :16-65, reduceResidualsNorm :67-80,
// getMaximumResidual :82-91)
bool residuals_to_remove(double quantile, size_t minimum_number_groups, Problem& problem, const IdMap& ids,
                         std::set<unsigned long>& to_remove) {
    if (ids.empty()) return true;
    std::map<unsigned long, double> max_res;
    for (const auto& el : ids) {
        double r[3], c;
        if (!problem.EvaluateBlock(*el.first, false, &c, r, nullptr)) {
            // Problem::Evaluate would fail as a whole; the reference ignores the return value and reads an empty
            // vector (undefined behaviour).  Treat the block as an infinitely large residual instead.
            r[0] = r[1] = r[2] = std::numeric_limits<double>::infinity();
        }
        double s = 0.0;
        for (int i = 0; i < el.second.second; ++i) s += r[i] * r[i];
        double n = std::sqrt(s);
        auto it = max_res.find(el.second.first);
        if (it == max_res.end())
            max_res[el.second.first] = n;
        else
            it->second = std::max(it->second, n);
    }
    if (max_res.size() < minimum_number_groups) return true;
    for (auto id : trim_quantile(max_res, quantile)) to_remove.insert(id);
    return true;
}

void print_summary(const SolverSummary& s) {
    std::printf("solve: term=%d '%s' e=%d f=%d res=%d fixed=%.6g\n", (int)s.termination, s.message.c_str(), s.num_e_blocks,
                s.num_f_params, s.num_residuals, s.fixed_cost);
    for (const auto& it : s.iterations)
        std::printf("  it %3d cost %.10e change %.3e |g| %.3e step %.3e rho %.3e radius %.3e valid %d succ %d\n", it.iteration,
                    it.cost, it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease,
                    it.trust_region_radius, (int)it.step_is_valid, (int)it.step_is_successful);
}

// group ids (= landmark indices of the caller's window) removed by the last solve_trimmed of this thread: lets a
// test rebuild the problem the final solve actually saw (tests/test_basin_restart.py)
static thread_local std::vector<unsigned long> g_last_removed;

struct TrimResult {
    std::vector<SolverSummary> summaries;
    int n_trimmed = 0;
    int extra_solves = 0, extra_iterations = 0, extra_successful = 0, extra_lin = 0;
};

// solveTrimmed, robust_solving.cpp:140-248.  trust_region_relaxation_factor is -10 on this path
// (bundle_adjuster_keyframes.cpp:763,885) => the radius is reset to the Ceres default before every solve.
TrimResult solve_trimmed(const std::vector<int>& number_iterations,
                         std::vector<std::pair<IdMap*, double>>& ids_trimmer_specs, Problem& problem,
                         SolverOptions options, size_t minimum_number_residual_groups) {
    TrimResult R;
    const int number_iterations_final = options.max_num_iterations;
    std::set<unsigned long> all_removed;
    for (int num_outlier_iter : number_iterations) {
        options.max_num_iterations = num_outlier_iter;
        SolverSummary cur;
        Solve(options, &problem, &cur);
        double cost_change = cur.initial_cost - cur.final_cost;
        if (cost_change <= 0.) {
            R.extra_solves += 1;  // the discarded first attempt still ran (reported in num_solves / iterations)
            R.extra_iterations += std::max(0, (int)cur.iterations.size() - 1);
            R.extra_successful += std::max(0, cur.num_successful_steps - 1);
            R.extra_lin += cur.num_successful_steps;
            options.max_num_iterations = 3 * num_outlier_iter;
            Solve(options, &problem, &cur);
        }
        R.summaries.push_back(cur);
        if (getenv("ORACLE_VERBOSE")) print_summary(cur);
        std::set<unsigned long> to_remove;
        for (auto& el : ids_trimmer_specs)
            residuals_to_remove(el.second, minimum_number_residual_groups, problem, *el.first, to_remove);
        for (auto& el : ids_trimmer_specs) {
            IdMap& m = *el.first;
            for (auto it = m.begin(); it != m.end();) {
                if (to_remove.count(it->second.first)) {
                    problem.RemoveResidualBlock(it->first);
                    it = m.erase(it);
                } else {
                    ++it;
                }
            }
        }
        problem.RemoveUnconstrainedParameters();
        for (auto id : to_remove) all_removed.insert(id);
    }
    options.max_num_iterations = number_iterations_final;
    SolverSummary fin;
    Solve(options, &problem, &fin);
    R.summaries.push_back(fin);
    if (getenv("ORACLE_VERBOSE")) print_summary(fin);
    R.n_trimmed = (int)all_removed.size();
    g_last_removed.assign(all_removed.begin(), all_removed.end());
    return R;
}

SolverOptions make_options(const limo_ba_options& o, int nt, int nlt) {
    SolverOptions s;
    s.max_num_iterations = o.max_num_iterations;
    s.max_solver_time_in_seconds = o.max_solver_time_sec;
    s.function_tolerance = o.function_tolerance;
    s.gradient_tolerance = o.gradient_tolerance;
    s.parameter_tolerance = o.parameter_tolerance;
    s.initial_trust_region_radius = o.initial_trust_region_radius;
    s.max_trust_region_radius = o.max_trust_region_radius;
    s.min_trust_region_radius = o.min_trust_region_radius;
    s.min_lm_diagonal = o.min_lm_diagonal;
    s.max_lm_diagonal = o.max_lm_diagonal;
    s.min_relative_decrease = o.min_relative_decrease;
    s.max_num_consecutive_invalid_steps = o.max_num_consecutive_invalid_steps;
    s.jacobi_scaling = o.jacobi_scaling != 0;
    s.num_threads = std::max(1, nt);
    s.num_linear_solver_threads = std::max(1, nlt);
    return s;
}

void fill_report(const TrimResult& R, const Built& B, limo_ba_report* rep, double* phase_times) {
    if (rep) {
        std::memset(rep, 0, sizeof(*rep));
        rep->termination = (int)R.summaries.back().termination;
        rep->num_solves = (int)R.summaries.size() + R.extra_solves;
        rep->iterations_total = R.extra_iterations;
        rep->successful_steps = R.extra_successful;
        rep->num_linearizations = R.extra_lin;
        for (const auto& s : R.summaries) {
            rep->iterations_total += std::max(0, (int)s.iterations.size() - 1);
            rep->successful_steps += std::max(0, s.num_successful_steps - 1);
            rep->num_linearizations += s.num_successful_steps;
        }
        rep->iterations_final = std::max(0, (int)R.summaries.back().iterations.size() - 1);
        rep->n_depth_blocks = B.n_depth;
        rep->n_repr_blocks = B.n_repr;
        rep->n_gp_blocks = B.n_gp;
        rep->n_trimmed_landmarks = R.n_trimmed;
        rep->initial_cost = R.summaries.front().initial_cost;
        rep->final_cost = R.summaries.back().final_cost;
    }
    if (phase_times) {
        phase_times[0] = phase_times[1] = phase_times[2] = phase_times[3] = 0;
        for (const auto& s : R.summaries) {
            phase_times[0] += s.time_eval;
            phase_times[1] += s.time_schur;
            phase_times[2] += s.time_chol;
            phase_times[3] += s.time_total;
        }
    }
}

// functors of robust_optimization/test/robust_optimization.cpp:108-132
struct RobustTestIn {
    template <typename T>
    bool operator()(const T* const p, T* r) const {
        r[0] = T(3.0) * p[0];
        return true;
    }
};
struct RobustTestOut {
    template <typename T>
    bool operator()(const T* const, T* r) const {
        r[0] = T(10.);
        return true;
    }
};

}  // namespace

extern "C" {

void oracle_ba_default_options(limo_ba_options* o) {
    o->depth_thres = 0.16;
    o->reprojection_thres = 1.6;
    o->depth_quantile = 0.95;
    o->reprojection_quantile = 0.95;
    o->num_trim_rounds = 1;
    o->trim_solver_iterations = 2;
    o->min_landmarks_for_trimming = 100;
    o->minimum_number_residual_groups = 30;
    o->max_num_iterations = 100;
    o->max_solver_time_sec = -1.0;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->min_relative_decrease = 1e-3;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
}

// BundleAdjusterKeyframes::solve() from "selection done" to "return report".
// phase_times (optional, 4 doubles): residual/Jacobian evaluation, Schur eliminate + back-substitute, Cholesky, total.
int oracle_ba_solve(limo_ba_window* w, const limo_ba_options* o, limo_ba_report* rep, int num_threads,
                    int num_linear_solver_threads, double* phase_times) {
    if (!w || !o) return LIMO_ERR_INVALID;
    if (w->n_kf < 1) return LIMO_ERR_NOT_ENOUGH_KF;  // (:630-632 counts ALL pushed keyframes: the caller's check; active ones may be 1 or 2)
    auto t0 = std::chrono::steady_clock::now();
    Built B;
    build_solve_problem(*w, *o, B);
    std::vector<int> number_iterations;
    if (w->n_lm > o->min_landmarks_for_trimming)  // :741-745
        for (int i = 0; i < o->num_trim_rounds; ++i) number_iterations.push_back(o->trim_solver_iterations);
    std::vector<std::pair<IdMap*, double>> input;  // :746-758
    input.push_back({&B.depth, o->depth_quantile});
    input.push_back({&B.repr, o->reprojection_quantile});
    input.push_back({&B.gp, 1.0});
    TrimResult R = solve_trimmed(number_iterations, input, B.problem, make_options(*o, num_threads, num_linear_solver_threads),
                                 (size_t)o->minimum_number_residual_groups);
    fill_report(R, B, rep, phase_times);
    if (rep) rep->time_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return LIMO_OK;
}

// Landmark indices trimmed by the last oracle_ba_solve / oracle_ba_adjust_pose_only on this thread; returns the count.
int oracle_last_trimmed(int32_t* out, int cap) {
    int n = 0;
    for (unsigned long id : g_last_removed) {
        if (n < cap && out) out[n] = (int32_t)id;
        ++n;
    }
    return n;
}

// adjustPoseOnly, :820-888.  window: n_kf == 1, landmarks constant.
int oracle_ba_adjust_pose_only(limo_ba_window* w, const limo_speed_prior* prior, const limo_ba_options* o,
                               limo_ba_report* rep, int num_threads) {
    if (!w || !o || w->n_kf != 1) return LIMO_ERR_INVALID;
    auto t0 = std::chrono::steady_clock::now();
    Built B;
    build_pose_only_problem(*w, prior, *o, B);
    std::vector<int> number_iterations;
    if (w->n_lm > o->min_landmarks_for_trimming)  // :865-869 (caller passes 30)
        for (int i = 0; i < o->num_trim_rounds; ++i) number_iterations.push_back(o->trim_solver_iterations);
    std::vector<std::pair<IdMap*, double>> input;
    input.push_back({&B.depth, o->depth_quantile});
    input.push_back({&B.repr, o->reprojection_quantile});
    TrimResult R = solve_trimmed(number_iterations, input, B.problem, make_options(*o, num_threads, 1),
                                 (size_t)o->minimum_number_residual_groups);
    fill_report(R, B, rep, nullptr);
    if (rep) rep->time_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return LIMO_OK;
}

// Problem::Evaluate over the reprojection/depth blocks, per observation (see limo_ba_evaluate).
int oracle_ba_evaluate(const limo_ba_window* w, const limo_ba_options* o, int apply_loss, double* cost,
                       double* residuals, double* jac_pose, double* jac_lm, uint8_t* valid) {
    if (!w || !o) return LIMO_ERR_INVALID;
    double total = 0.0;
    for (int i = 0; i < w->n_obs; ++i) {
        const int k = w->obs_kf[i], l = w->obs_lm[i], c = w->obs_cam[i];
        const double* cam = w->cam + 10 * c;
        ParamBlock pose, lm;
        pose.user = w->kf_pose + 7 * k;
        pose.size = 7;
        pose.kind = PK_POSE_QUAT_R3;
        lm.user = w->lm_pos + 3 * l;
        lm.size = 3;
        double r3[3] = {0, 0, 0}, jp[18], jl[9];
        for (int q = 0; q < 18; ++q) jp[q] = 0;
        for (int q = 0; q < 9; ++q) jl[q] = 0;
        bool ok = true;
        Problem P;
        {
            ResidualBlock rb;
            ReprojectionErrorWithQuaternions f{static_cast<double>(w->obs_u[i]), static_cast<double>(w->obs_v[i]),
                                               cam[0], cam[1], cam[2], cam_pose(cam)};
            rb.cost.reset(new AutoDiffCost<ReprojectionErrorWithQuaternions, 2, 7, 3>(f));
            rb.loss = Loss::ScaledCauchy(o->reprojection_thres, w->lm_weight[l]);
            rb.params = {&pose, &lm};
            double* jj[4] = {jp, jl, nullptr, nullptr};
            double cst;
            ok = P.EvaluateBlock(rb, apply_loss != 0, &cst, r3, jj);
            if (ok) total += cst;
        }
        if (ok && w->obs_d[i] > 0.0f) {
            ResidualBlock rb;
            LandmarkDepthError f{static_cast<double>(w->obs_d[i]), cam_pose(cam)};
            rb.cost.reset(new AutoDiffCost<LandmarkDepthError, 1, 7, 3>(f));
            rb.loss = Loss::ScaledCauchy(o->depth_thres, w->lm_weight[l]);
            rb.params = {&pose, &lm};
            double* jj[4] = {jp + 12, jl + 6, nullptr, nullptr};
            double cst;
            bool ok2 = P.EvaluateBlock(rb, apply_loss != 0, &cst, r3 + 2, jj);
            if (ok2) total += cst;
        }
        if (!ok) {
            r3[0] = r3[1] = r3[2] = 0;
            for (int q = 0; q < 18; ++q) jp[q] = 0;
            for (int q = 0; q < 9; ++q) jl[q] = 0;
        }
        if (valid) valid[i] = ok ? 1 : 0;
        if (residuals) std::memcpy(residuals + 3 * (size_t)i, r3, sizeof(r3));
        if (jac_pose) std::memcpy(jac_pose + 18 * (size_t)i, jp, sizeof(jp));
        if (jac_lm) std::memcpy(jac_lm + 9 * (size_t)i, jl, sizeof(jl));
    }
    if (cost) *cost = total;
    return LIMO_OK;
}

// The residual blocks of the solve() / adjustPoseOnly problem that are not reprojection / depth blocks, row by row, with
// their tangent-space Jacobians by dual numbers (see limo_ba_evaluate_rows in include/limo_hip.h: same row struct, same
// conventions; the ORDER of the rows is this file's construction order - match rows by (kind, kf, lm, sub)).
int oracle_ba_evaluate_rows(const limo_ba_window* w, const limo_speed_prior* prior, int pose_only, const limo_ba_options* o, int32_t cap,
                            limo_ba_row* rows, int32_t* n_rows) {
    if (!w || !o || !n_rows) return LIMO_ERR_INVALID;
    Built B;
    if (pose_only)
        build_pose_only_problem(*w, prior, *o, B);
    else
        build_solve_problem(*w, *o, B);
    int n = 0;
    for (const RowMeta& m : B.other) {
        const ResidualBlock& rb = *m.rb;
        const int nres = rb.cost->nres, np = (int)rb.params.size();
        double res[3], cost, jac[4][3 * 7];
        double* jj[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < np; ++i) jj[i] = jac[i];
        if (!B.problem.EvaluateBlock(rb, true, &cost, res, jj)) return 1;
        // keyframes of the block's parameters
        int kfs[2] = {-1, -1};
        bool all_const = true;
        auto kf_of = [&](const ParamBlock* p, int& slot0) {
            const double* u = p->user;
            if (u >= w->kf_pose && u < w->kf_pose + 7 * (size_t)w->n_kf) {
                slot0 = 0;
                return (int)((u - w->kf_pose) / 7);
            }
            if (u >= w->kf_plane_dir && u < w->kf_plane_dir + 3 * (size_t)w->n_kf) {
                slot0 = 6;
                return (int)((u - w->kf_plane_dir) / 3);
            }
            if (u >= w->kf_plane_dist && u < w->kf_plane_dist + w->n_kf) {
                slot0 = 9;
                return (int)(u - w->kf_plane_dist);
            }
            slot0 = -1;  // a landmark
            return -1;
        };
        for (int i = 0; i < np; ++i) {
            int s0;
            const int k = kf_of(rb.params[i], s0);
            all_const = all_const && rb.params[i]->constant;
            if (k < 0) continue;
            if (kfs[0] < 0 || k == kfs[0])
                kfs[0] = k;
            else
                kfs[1] = k;
        }
        if (kfs[1] >= 0 && kfs[1] < kfs[0]) std::swap(kfs[0], kfs[1]);
        for (int r = 0; r < nres; ++r, ++n) {
            if (n >= cap || !rows) continue;
            limo_ba_row& out = rows[n];
            std::memset(&out, 0, sizeof(out));
            out.kind = m.kind;
            out.sub = r;
            out.kf[0] = kfs[0];
            out.kf[1] = kfs[1];
            out.lm = m.lm;
            out.fixed = all_const ? 1 : 0;
            out.r = res[r];
            out.cost = r == 0 ? cost : 0.0;
            for (int i = 0; i < np; ++i) {
                int s0;
                const int k = kf_of(rb.params[i], s0);
                const int l = rb.params[i]->lsize();
                for (int c = 0; c < l; ++c) {
                    if (k < 0)
                        out.jac_lm[c] += jac[i][r * l + c];
                    else
                        out.jac_kf[k == kfs[0] ? 0 : 1][s0 + c] += jac[i][r * l + c];
                }
            }
        }
    }
    *n_rows = n;
    return LIMO_OK;
}

// Total cost (with loss) and block counts of the solve() problem at the window's current parameters.
int oracle_ba_problem_cost(const limo_ba_window* w, const limo_ba_options* o, double* cost, int32_t* counts3) {
    if (!w || !o) return LIMO_ERR_INVALID;
    Built B;
    build_solve_problem(*w, *o, B);
    std::vector<ResidualBlock*> all;
    for (auto& b : B.problem.blocks) all.push_back(b.get());
    double c = 0.0;
    bool ok = B.problem.Evaluate(all, true, &c, nullptr);
    if (cost) *cost = c;
    if (counts3) {
        counts3[0] = B.n_depth;
        counts3[1] = B.n_repr;
        counts3[2] = B.n_gp;
    }
    return ok ? LIMO_OK : 1;
}

int oracle_trim_quantile(int32_t n, const int64_t* ids, const double* values, double quantile, int64_t* out) {
    std::map<unsigned long, double> m;
    for (int i = 0; i < n; ++i) m[(unsigned long)ids[i]] = values[i];
    auto v = trim_quantile(m, quantile);
    for (size_t i = 0; i < v.size(); ++i) out[i] = (int64_t)v[i];
    return (int)v.size();
}

// TrimmerFix::getOutliers, trimmer_fix.hpp:38-46
int oracle_trim_fix(int32_t n, const int64_t* ids, const double* values, double thres, int64_t* out) {
    int k = 0;
    for (int i = 0; i < n; ++i)
        if (values[i] > thres) out[k++] = ids[i];
    return k;
}

// Known-answer access to single functors (tests/golden): kind selects the functor, consts its constructor
// arguments, p0..p3 the parameter blocks; residuals receives kNumResiduals values.  Returns 1 if the functor
// returned true, 0 if false, <0 on bad kind.
int oracle_functor(int kind, const double* consts, const double* p0, const double* p1, const double* p2,
                   const double* p3, double* residuals) {
    switch (kind) {
        case 0: {  // ReprojectionErrorWithQuaternions: consts = u, v, f, cx, cy, pose_C_X[7]
            ReprojectionErrorWithQuaternions f{consts[0], consts[1], consts[2], consts[3], consts[4], {}};
            for (int i = 0; i < 7; ++i) f.pose_C_X[i] = consts[5 + i];
            return f(p0, p1, residuals) ? 1 : 0;
        }
        case 1: {  // LandmarkDepthError: consts = d, pose_C_X[7]
            LandmarkDepthError f{consts[0], {}};
            for (int i = 0; i < 7; ++i) f.pose_C_X_[i] = consts[1 + i];
            return f(p0, p1, residuals) ? 1 : 0;
        }
        case 2: {
            PoseRegularization f{consts[0]};
            return f(p0, p1, residuals) ? 1 : 0;
        }
        case 3: {
            GroundPlaneHeightRegularization f;
            return f(p0, p1, p2, p3, residuals) ? 1 : 0;
        }
        case 4: {
            GroundPlaneMotionRegularization f;
            return f(p0, p1, p2, residuals) ? 1 : 0;
        }
        case 5: {
            TranslationDifferenceRegularization f;
            return f(p0, p1, p2, residuals) ? 1 : 0;
        }
        case 6: {
            VectorDifferenceRegularization f;
            return f(p0, p1, residuals) ? 1 : 0;
        }
        case 7: {
            VectorDifferenceRegularization2 f{{consts[0], consts[1], consts[2]}};
            return f(p0, residuals) ? 1 : 0;
        }
        case 8: {
            GroundPlaneDistanceRegularization f;
            return f(p0, p1, residuals) ? 1 : 0;
        }
        case 9: {  // SpeedRegularizationVector2: consts = ts_cur, ts_before, ts_before2, pose_before[7], pose_before2[7]
            SpeedRegularizationVector2 f;
            Pose7 a, b;
            for (int i = 0; i < 7; ++i) {
                a[i] = consts[3 + i];
                b[i] = consts[10 + i];
            }
            if (!SpeedRegularizationVector2::make(consts[0], consts[1], consts[2], a, b, f)) return -2;
            return f(p0, residuals) ? 1 : 0;
        }
        case 10: {  // MotionModelRegularization (internal/motion_model_regularization.hpp:32-74): NOT on the solve path
                    // (included at bundle_adjuster_keyframes.cpp:20, never instantiated); restated for its known-answer
                    // test only (test/keyframe_bundle_adjustment.cpp:1212-1276), plain doubles
            const Iso<double> p0_ = convert(p1), p1_ = convert(p0);  // operator()(pose_keyframe1_origin, pose_keyframe0_origin)
            const Iso<double> m = compose(p1_, inverse(p0_));
            // Eigen::Quaternion(rotation matrix) (Shepperd), then x = y = 0, normalise, yaw = sign(z) 2 acos(w)
            double q[4];
            const double* R = m.R;
            double t = R[0] + R[4] + R[8];
            if (t > 0.0) {
                t = std::sqrt(t + 1.0);
                q[0] = 0.5 * t;
                t = 0.5 / t;
                q[1] = (R[7] - R[5]) * t;
                q[2] = (R[2] - R[6]) * t;
                q[3] = (R[3] - R[1]) * t;
            } else {
                int i = 0;
                if (R[4] > R[0]) i = 1;
                if (R[8] > R[4 * i]) i = 2;
                const int j = (i + 1) % 3, k = (j + 1) % 3;
                t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
                q[1 + i] = 0.5 * t;
                t = 0.5 / t;
                q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
                q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
                q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
            }
            const double n = std::sqrt(q[0] * q[0] + q[3] * q[3]);
            const double w = q[0] / n, z = q[3] / n;
            const double yaw = (z < 0.0 ? -1.0 : 1.0) * 2.0 * std::acos(w);
            const double dy = std::fabs(yaw) < 1.e-6 ? 0.0 : m.t[0] / std::sin(yaw) * (1.0 - std::cos(yaw));
            residuals[0] = m.t[1] - dy;
            residuals[1] = m.t[2];
            return 1;
        }
    }
    return -1;
}

// Ambient Jacobians of the functors of oracle_functor kinds 0 (reprojection), 1 (depth), 3 (ground height) by dual
// numbers, for the sympy cross-check (tests/test_oracle_crosscheck.py): jac = concatenation over the parameter blocks
// of row-major nres x size matrices.  Returns nres, 0 if the functor returned false, <0 on a bad kind.
int oracle_functor_jacobian(int kind, const double* consts, const double* p0, const double* p1, const double* p2,
                            const double* p3, double* residuals, double* jac) {
    const double* ps[4] = {p0, p1, p2, p3};
    std::unique_ptr<CostFunction> cost;
    if (kind == 0) {
        ReprojectionErrorWithQuaternions f{consts[0], consts[1], consts[2], consts[3], consts[4], {}};
        for (int i = 0; i < 7; ++i) f.pose_C_X[i] = consts[5 + i];
        cost.reset(new AutoDiffCost<ReprojectionErrorWithQuaternions, 2, 7, 3>(f));
    } else if (kind == 1) {
        LandmarkDepthError f{consts[0], {}};
        for (int i = 0; i < 7; ++i) f.pose_C_X_[i] = consts[1 + i];
        cost.reset(new AutoDiffCost<LandmarkDepthError, 1, 7, 3>(f));
    } else if (kind == 3) {
        cost.reset(new AutoDiffCost<GroundPlaneHeightRegularization, 1, 7, 3, 1, 3>(GroundPlaneHeightRegularization()));
    } else {
        return -1;
    }
    double* jj[4] = {nullptr, nullptr, nullptr, nullptr};
    int off = 0;
    for (size_t i = 0; i < cost->sizes.size(); ++i) {
        jj[i] = jac + off;
        off += cost->nres * cost->sizes[i];
    }
    return cost->Evaluate(ps, residuals, jj) ? cost->nres : 0;
}

// First trust-region step of the solve() problem of `w` (w is left at its input values): fills the probe arrays (see
// StepProbe).  sizes3 = (num_residuals, num_eff, num_e); arrays may be null to query the sizes first.
int oracle_ba_first_step(const limo_ba_window* w, const limo_ba_options* o, int32_t* sizes3, double* J, double* r, double* D, double* y) {
    if (!w || !o || !sizes3) return LIMO_ERR_INVALID;
    // private copies of the parameter arrays: Solve() writes the result of its one iteration back
    std::vector<double> pose(w->kf_pose, w->kf_pose + 7 * (size_t)w->n_kf), dir(w->kf_plane_dir, w->kf_plane_dir + 3 * (size_t)w->n_kf),
        dist(w->kf_plane_dist, w->kf_plane_dist + w->n_kf), lm(w->lm_pos, w->lm_pos + 3 * (size_t)w->n_lm);
    limo_ba_window c = *w;
    c.kf_pose = pose.data();
    c.kf_plane_dir = dir.data();
    c.kf_plane_dist = dist.data();
    c.lm_pos = lm.data();
    Built B;
    build_solve_problem(c, *o, B);
    SolverOptions so = make_options(*o, 1, 1);
    so.max_num_iterations = 1;
    StepProbe probe;
    so.probe = &probe;
    SolverSummary sum;
    Solve(so, &B.problem, &sum);
    if (!probe.filled || probe.y.empty()) return 1;
    sizes3[0] = probe.num_residuals;
    sizes3[1] = probe.num_eff;
    sizes3[2] = probe.num_e;
    if (J) std::memcpy(J, probe.J.data(), sizeof(double) * probe.J.size());
    if (r) std::memcpy(r, probe.r.data(), sizeof(double) * probe.r.size());
    if (D) std::memcpy(D, probe.D.data(), sizeof(double) * probe.D.size());
    if (y) std::memcpy(y, probe.y.data(), sizeof(double) * probe.y.size());
    return LIMO_OK;
}

// loss function known-answer access: kind 0 trivial, 1 huber, 2 cauchy; scaled by weight
void oracle_loss(int kind, double a, double weight, double s, double* rho3) {
    Loss l;
    l.kind = kind == 1 ? Loss::HUBER : kind == 2 ? Loss::CAUCHY : Loss::TRIVIAL;
    l.a = a;
    l.scaled = true;
    l.weight = weight;
    l.Evaluate(s, rho3);
}

// local parameterisations: kind 0 = pose (quaternion x R3), 1 = FixScaleVectorPlus
void oracle_plus(int kind, const double* x, const double* delta, double* out, double* jac) {
    ParamBlock p;
    p.kind = kind == 0 ? PK_POSE_QUAT_R3 : PK_FIX_SCALE_VECTOR;
    p.size = kind == 0 ? 7 : 3;
    if (out) p.Plus(x, delta, out);
    if (jac) p.ComputeJacobian(x, jac);
}

// Generic trimmed least squares used by the reference's own robust_optimization test
// (robust_optimization/test/robust_optimization.cpp:134-156): n_in residuals 3*x, n_out residuals const 10.
double oracle_robust_test_solve_trimmed(int n_in, int n_out, double x0, const int* schedule, int n_sched, double q) {
    double x = x0;
    Problem P;
    ParamBlock* pb = P.AddParameterBlock(&x, 1);
    IdMap ids;
    unsigned long g = 0;
    for (int i = 0; i < n_in; ++i) ids[P.AddResidualBlock(new AutoDiffCost<RobustTestIn, 1, 1>(RobustTestIn()), Loss::None(), {pb})] = {g++, 1};
    for (int i = 0; i < n_out; ++i) ids[P.AddResidualBlock(new AutoDiffCost<RobustTestOut, 1, 1>(RobustTestOut()), Loss::None(), {pb})] = {g++, 1};
    std::vector<int> sched(schedule, schedule + n_sched);
    std::vector<std::pair<IdMap*, double>> input{{&ids, q}};
    SolverOptions so;  // getStandardSolverOptions(0.1): DENSE_SCHUR, 100 iterations (time cap disabled here)
    solve_trimmed(sched, input, P, so, 30);
    return x;
}

int oracle_num_procs(void) {
    return omp_get_num_procs();
}

}  // extern "C"
