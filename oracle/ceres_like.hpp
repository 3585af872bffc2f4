// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// ceres_like.hpp — CPU restatement of the parts of Ceres Solver the reference drives
// (third-party dependency ABSENT from /root/reference: "libceres-dev" of Ubuntu 18.04 = Ceres 1.13.x with
// Eigen 3.3.4, pinned only by docker/src/Dockerfile:14,47).  Restated from Ceres' published algorithm
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, schur_eliminator_impl.h, corrector.cc,
// loss_function.cc, local_parameterization.cc, residual_block.cc of the 1.13 release) and anchored on the
// reference's call sites:
//   ceres::Solve                     robust_optimization/src/robust_solving.cpp:169,174,239
//   Problem::Evaluate                robust_solving.cpp:44, keyframe_bundle_adjustment/src/definitions.cpp:94
//   AddResidualBlock                 bundle_adjuster_keyframes.cpp:551,587,614,779,786,794,811,849,899
//   RemoveResidualBlock / RemoveParameterBlock   robust_solving.cpp:208,134
//   options                          robust_optimization/include/robust_optimization/robust_solving.hpp:93-108
// PARITY UNPINNED at the iteration level: no Ceres binary exists here to compare against; the reference's own
// tests only pin convergence to ground truth (see tests/golden and oracle/README.md).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "functors.hpp"
#include "jet.hpp"

namespace kba_oracle {

// ----------------------------------------------------------------------------- loss functions (loss_function.cc)
struct Loss {
    enum Kind { NONE, TRIVIAL, HUBER, CAUCHY } kind = NONE;
    double a = 1.0;       // scale of Huber / Cauchy
    bool scaled = false;  // wrapped in ScaledLoss(inner, weight)
    double weight = 1.0;
    static Loss None() { return Loss(); }
    static Loss ScaledTrivial(double w) {
        Loss l;
        l.kind = TRIVIAL;
        l.scaled = true;
        l.weight = w;
        return l;
    }
    static Loss ScaledHuber(double a, double w) {
        Loss l;
        l.kind = HUBER;
        l.a = a;
        l.scaled = true;
        l.weight = w;
        return l;
    }
    static Loss ScaledCauchy(double a, double w) {
        Loss l;
        l.kind = CAUCHY;
        l.a = a;
        l.scaled = true;
        l.weight = w;
        return l;
    }
    void Evaluate(double s, double rho[3]) const {
        switch (kind) {
            case NONE:
            case TRIVIAL:
                rho[0] = s;
                rho[1] = 1.0;
                rho[2] = 0.0;
                break;
            case HUBER: {
                const double b = a * a;
                if (s > b) {
                    const double r = std::sqrt(s);
                    rho[0] = 2.0 * a * r - b;
                    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
                    rho[2] = -rho[1] / (2.0 * s);
                } else {
                    rho[0] = s;
                    rho[1] = 1.0;
                    rho[2] = 0.0;
                }
                break;
            }
            case CAUCHY: {
                const double b = a * a;
                const double c = 1.0 / b;
                const double sum = 1.0 + s * c;
                const double inv = 1.0 / sum;
                rho[0] = b * std::log(sum);
                rho[1] = std::max(std::numeric_limits<double>::min(), inv);
                rho[2] = -c * (inv * inv);
                break;
            }
        }
        if (scaled) {
            rho[0] *= weight;
            rho[1] *= weight;
            rho[2] *= weight;
        }
    }
};

// ----------------------------------------------------------------------------- local parameterisations
enum ParamKind { PK_EUCLIDEAN, PK_POSE_QUAT_R3, PK_FIX_SCALE_VECTOR };

struct ParamBlock {
    double* user = nullptr;  // user memory (optimised in place at the end of Solve)
    int size = 0;
    ParamKind kind = PK_EUCLIDEAN;
    bool constant = false;
    bool is_landmark = false;  // candidate for the Schur-eliminated independent set
    int nres = 0;              // number of (not removed) residual blocks referencing it
    // program bookkeeping (valid inside one Solve)
    int state_off = -1;
    int delta_off = -1;
    int e_index = -1;
    const double* state = nullptr;  // pointer the cost functions read from

    int lsize() const { return kind == PK_POSE_QUAT_R3 ? 6 : size; }

    // x_plus_delta = Plus(x, delta)
    bool Plus(const double* x, const double* delta, double* out) const {
        switch (kind) {
            case PK_EUCLIDEAN:
                for (int i = 0; i < size; ++i) out[i] = x[i] + delta[i];
                return true;
            case PK_POSE_QUAT_R3: {
                // ProductParameterization(QuaternionParameterization, IdentityParameterization(3)),
                // bundle_adjuster_keyframes.cpp:181-182.  QuaternionParameterization::Plus (local_parameterization.cc)
                const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
                if (norm_delta > 0.0) {
                    const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
                    double q_delta[4] = {std::cos(norm_delta), sin_delta_by_delta * delta[0],
                                         sin_delta_by_delta * delta[1], sin_delta_by_delta * delta[2]};
                    // QuaternionProduct(q_delta, x, x_plus_delta)  (rotation.h)
                    const double* z = q_delta;
                    const double* w = x;
                    out[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
                    out[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
                    out[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
                    out[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
                } else {
                    for (int i = 0; i < 4; ++i) out[i] = x[i];
                }
                for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + delta[3 + i];
                return true;
            }
            case PK_FIX_SCALE_VECTOR: {
                FixScaleVectorPlus f;
                return f(x, delta, out);
            }
        }
        return false;
    }
    // jac = d Plus(x, delta) / d delta at delta = 0, row-major size x lsize
    void ComputeJacobian(const double* x, double* jac) const {
        switch (kind) {
            case PK_EUCLIDEAN:
                for (int i = 0; i < size * size; ++i) jac[i] = 0.0;
                for (int i = 0; i < size; ++i) jac[i * size + i] = 1.0;
                return;
            case PK_POSE_QUAT_R3: {
                for (int i = 0; i < 42; ++i) jac[i] = 0.0;
                // QuaternionParameterization::ComputeJacobian
                const double q[12] = {-x[1], -x[2], -x[3], x[0], x[3], -x[2], -x[3], x[0], x[1], x[2], -x[1], x[0]};
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 3; ++c) jac[r * 6 + c] = q[r * 3 + c];
                for (int i = 0; i < 3; ++i) jac[(4 + i) * 6 + 3 + i] = 1.0;
                return;
            }
            case PK_FIX_SCALE_VECTOR: {
                // AutoDiffLocalParameterization<FixScaleVectorPlus,3,3>, bundle_adjuster_keyframes.cpp:190-193
                using J = Jet<3>;
                J xx[3] = {J(x[0]), J(x[1]), J(x[2])};
                J dd[3] = {J(0.0, 0), J(0.0, 1), J(0.0, 2)};
                J out[3];
                FixScaleVectorPlus f;
                f(xx, dd, out);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) jac[r * 3 + c] = out[r].v[c];
                return;
            }
        }
    }
};

// ----------------------------------------------------------------------------- cost functions (AutoDiffCostFunction)
struct CostFunction {
    int nres = 0;
    std::vector<int> sizes;
    virtual ~CostFunction() {}
    // jacobians[i]: row-major nres x sizes[i], or nullptr
    virtual bool Evaluate(double const* const* params, double* residuals, double** jacobians) const = 0;
};

template <typename Functor, int kRes, int N0, int N1 = 0, int N2 = 0, int N3 = 0>
struct AutoDiffCost : CostFunction {
    Functor f;
    static constexpr int kN = N0 + N1 + N2 + N3;
    explicit AutoDiffCost(const Functor& fun) : f(fun) {
        nres = kRes;
        sizes.push_back(N0);
        if (N1) sizes.push_back(N1);
        if (N2) sizes.push_back(N2);
        if (N3) sizes.push_back(N3);
    }
    template <typename T>
    bool call(const T* p0, const T* p1, const T* p2, const T* p3, T* r) const {
        if constexpr (N3 > 0) {
            return f(p0, p1, p2, p3, r);
        } else if constexpr (N2 > 0) {
            (void)p3;
            return f(p0, p1, p2, r);
        } else if constexpr (N1 > 0) {
            (void)p2;
            (void)p3;
            return f(p0, p1, r);
        } else {
            (void)p1;
            (void)p2;
            (void)p3;
            return f(p0, r);
        }
    }
    bool Evaluate(double const* const* params, double* residuals, double** jacobians) const override {
        if (!jacobians) {
            return call<double>(params[0], N1 ? params[1] : nullptr, N2 ? params[2] : nullptr,
                                N3 ? params[3] : nullptr, residuals);
        }
        using J = Jet<kN>;
        J x[kN];
        const int ns[4] = {N0, N1, N2, N3};
        int off = 0;
        for (int b = 0; b < 4; ++b) {
            for (int i = 0; i < ns[b]; ++i) x[off + i] = J(params[b][i], off + i);
            off += ns[b];
        }
        J r[kRes];
        bool ok = call<J>(x, x + N0, x + N0 + N1, x + N0 + N1 + N2, r);
        if (!ok) return false;
        for (int i = 0; i < kRes; ++i) residuals[i] = r[i].a;
        off = 0;
        for (int b = 0; b < 4; ++b) {
            if (ns[b] && jacobians[b]) {
                for (int i = 0; i < kRes; ++i)
                    for (int c = 0; c < ns[b]; ++c) jacobians[b][i * ns[b] + c] = r[i].v[off + c];
            }
            off += ns[b];
        }
        return true;
    }
};

struct ResidualBlock {
    std::unique_ptr<CostFunction> cost;
    Loss loss;
    std::vector<ParamBlock*> params;
    bool removed = false;
    // program bookkeeping
    int res_off = -1;  // offset into residual vector
    int jac_off = -1;  // offset into jacobian storage
};

struct IterationSummary {
    int iteration = 0;
    bool step_is_valid = false;
    bool step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, gradient_norm = 0, step_norm = 0, relative_decrease = 0,
           trust_region_radius = 0;
};

enum Termination { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };

struct SolverSummary {
    Termination termination = NO_CONVERGENCE;
    std::string message;
    double initial_cost = -1, final_cost = -1, fixed_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    std::vector<IterationSummary> iterations;
    int num_e_blocks = 0, num_f_params = 0, num_residuals = 0;
    double time_eval = 0, time_schur = 0, time_chol = 0, time_total = 0;
};

// Test probe (tests/test_oracle_crosscheck.py): the linear least-squares problem of the FIRST trust-region step as the
// Schur-based LinearSolve saw it - dense (column-scaled) Jacobian, residuals, LM diagonal - and the step it returned, so
// that a dense solver outside this library can solve  min |J y - r|^2 + |D y|^2  independently.
struct StepProbe {
    bool filled = false;
    int num_residuals = 0, num_eff = 0, num_e = 0;
    std::vector<double> J, r, D, y;  // J row-major num_residuals x num_eff; e-blocks (landmarks) first
};

struct SolverOptions {  // ceres::Solver::Options defaults (1.13) + the reference's overrides
    StepProbe* probe = nullptr;
    int max_num_iterations = 100;
    double max_solver_time_in_seconds = -1;  // <= 0: no wall-clock stop
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
    double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32, min_relative_decrease = 1e-3;
    int max_num_consecutive_invalid_steps = 5;
    bool jacobi_scaling = true;
    int num_threads = 1;                // residual/Jacobian evaluation
    int num_linear_solver_threads = 1;  // Schur elimination
};

class Problem {
public:
    std::vector<std::unique_ptr<ParamBlock>> params;
    std::map<const double*, ParamBlock*> by_ptr;
    std::vector<std::unique_ptr<ResidualBlock>> blocks;

    ParamBlock* AddParameterBlock(double* x, int size, ParamKind kind = PK_EUCLIDEAN, bool is_landmark = false) {
        auto it = by_ptr.find(x);
        if (it != by_ptr.end()) return it->second;
        params.emplace_back(new ParamBlock());
        ParamBlock* p = params.back().get();
        p->user = x;
        p->size = size;
        p->kind = kind;
        p->is_landmark = is_landmark;
        by_ptr[x] = p;
        return p;
    }
    bool HasParameterBlock(const double* x) const { return by_ptr.count(x) > 0; }
    ParamBlock* Get(const double* x) const {
        auto it = by_ptr.find(x);
        return it == by_ptr.end() ? nullptr : it->second;
    }
    ResidualBlock* AddResidualBlock(CostFunction* cost, const Loss& loss, const std::vector<ParamBlock*>& ps) {
        blocks.emplace_back(new ResidualBlock());
        ResidualBlock* b = blocks.back().get();
        b->cost.reset(cost);
        b->loss = loss;
        b->params = ps;
        for (auto* p : ps) p->nres++;
        return b;
    }
    void RemoveResidualBlock(ResidualBlock* b) {
        if (b->removed) return;
        b->removed = true;
        for (auto* p : b->params) p->nres--;
    }
    // robust_solving.cpp:127-137: parameter blocks without residual blocks leave the problem.
    void RemoveUnconstrainedParameters() {
        for (auto it = by_ptr.begin(); it != by_ptr.end();) {
            if (it->second->nres == 0)
                it = by_ptr.erase(it);
            else
                ++it;
        }
    }

    // ResidualBlock::Evaluate (residual_block.cc): cost function -> local parameterisation -> loss -> corrector.
    // jac_local[i]: row-major nres x lsize_i for non-constant params (nullptr entries skipped).
    // Parameter values are read from ParamBlock::state when set, else from user memory.
    bool EvaluateBlock(const ResidualBlock& b, bool apply_loss, double* cost, double* residuals,
                       double** jac_local) const {
        const int np = (int)b.params.size();
        const double* ps[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < np; ++i) ps[i] = b.params[i]->state ? b.params[i]->state : b.params[i]->user;
        const int nres = b.cost->nres;
        double jac_amb_store[64];
        double* jac_amb[4] = {nullptr, nullptr, nullptr, nullptr};
        if (jac_local) {
            int off = 0;
            for (int i = 0; i < np; ++i) {
                if (jac_local[i]) {
                    jac_amb[i] = jac_amb_store + off;
                    off += nres * b.params[i]->size;
                }
            }
        }
        if (!b.cost->Evaluate(ps, residuals, jac_local ? jac_amb : nullptr)) return false;
        if (jac_local) {
            for (int i = 0; i < np; ++i) {
                if (!jac_local[i]) continue;
                const ParamBlock* p = b.params[i];
                const int n = p->size, l = p->lsize();
                if (p->kind == PK_EUCLIDEAN) {
                    std::memcpy(jac_local[i], jac_amb[i], sizeof(double) * nres * n);
                } else {
                    double P[7 * 6];
                    p->ComputeJacobian(ps[i], P);
                    for (int r = 0; r < nres; ++r)
                        for (int c = 0; c < l; ++c) {
                            double s = 0.0;
                            for (int k = 0; k < n; ++k) s += jac_amb[i][r * n + k] * P[k * l + c];
                            jac_local[i][r * l + c] = s;
                        }
                }
            }
        }
        double sq = 0.0;
        for (int r = 0; r < nres; ++r) sq += residuals[r] * residuals[r];
        if (b.loss.kind == Loss::NONE || !apply_loss) {
            *cost = 0.5 * sq;
            return true;
        }
        double rho[3] = {0.0, 0.0, 0.0};
        b.loss.Evaluate(sq, rho);
        *cost = 0.5 * rho[0];
        // Corrector (corrector.cc): every loss on this path has rho'' <= 0 -> plain sqrt(rho') scaling.
        double residual_scaling, alpha_sq_norm;
        const double sqrt_rho1 = std::sqrt(rho[1]);
        if (sq == 0.0 || rho[2] <= 0.0) {
            residual_scaling = sqrt_rho1;
            alpha_sq_norm = 0.0;
        } else {
            const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
            const double alpha = 1.0 - std::sqrt(D);
            residual_scaling = sqrt_rho1 / (1 - alpha);
            alpha_sq_norm = alpha / sq;
        }
        if (jac_local) {
            for (int i = 0; i < np; ++i) {
                if (!jac_local[i]) continue;
                const int l = b.params[i]->lsize();
                if (alpha_sq_norm == 0.0) {
                    for (int k = 0; k < nres * l; ++k) jac_local[i][k] *= sqrt_rho1;
                } else {
                    // J = sqrt_rho1 * (J - alpha_sq_norm * r r^T J)
                    for (int c = 0; c < l; ++c) {
                        double rtj = 0.0;
                        for (int r = 0; r < nres; ++r) rtj += jac_local[i][r * l + c] * residuals[r];
                        for (int r = 0; r < nres; ++r)
                            jac_local[i][r * l + c] =
                                sqrt_rho1 * (jac_local[i][r * l + c] - alpha_sq_norm * residuals[r] * rtj);
                    }
                }
            }
        }
        for (int r = 0; r < nres; ++r) residuals[r] *= residual_scaling;
        return true;
    }

    // Problem::Evaluate for a list of residual blocks (robust_solving.cpp:24-44): returns cost and the
    // concatenated residuals.  Returns false if any block fails to evaluate.
    bool Evaluate(const std::vector<ResidualBlock*>& list, bool apply_loss, double* cost,
                  std::vector<double>* residuals) const {
        double total = 0.0;
        if (residuals) residuals->clear();
        for (auto* b : list) {
            double r[3], c;
            if (!EvaluateBlock(*b, apply_loss, &c, r, nullptr)) return false;
            total += c;
            if (residuals)
                for (int i = 0; i < b->cost->nres; ++i) residuals->push_back(r[i]);
        }
        if (cost) *cost = total;
        return true;
    }
};

void Solve(const SolverOptions& options, Problem* problem, SolverSummary* summary);

// small dense helpers shared with the solver / tests
bool CholeskyUpperInPlace(double* A, int n);  // A = U^T U, upper triangle in row-major full storage
void CholeskySolveUpper(const double* U, int n, double* b);
bool InvertPSD3(const double* m, double* inv);

}  // namespace kba_oracle
