// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// ceres_like.cpp — restated Ceres 1.13 trust-region minimiser (Levenberg-Marquardt, DENSE_SCHUR).
// Follows the published control flow of TrustRegionMinimizer::Minimize (trust_region_minimizer.cc):
//   IterationZero -> loop { ComputeTrustRegionStep; invalid? ; ComputeCandidatePointAndEvaluateCost;
//   ParameterToleranceReached; FunctionToleranceReached; IsStepSuccessful ? HandleSuccessfulStep : HandleUnsuccessfulStep }
// and LevenbergMarquardtStrategy::{ComputeStep,StepAccepted,StepRejected,StepIsInvalid}, SchurEliminator::{Eliminate,
// BackSubstitute}, DenseSchurComplementSolver::SolveReducedLinearSystem (Eigen LLT).  Call sites in the reference:
// robust_optimization/src/robust_solving.cpp:169,174,239.
#include "ceres_like.hpp"

#include <omp.h>

#include <cstdio>

namespace kba_oracle {

bool CholeskyUpperInPlace(double* A, int n) {
    // Eigen LLT<Upper> semantics: fails when a pivot is <= 0.
    for (int k = 0; k < n; ++k) {
        double d = A[k * n + k];
        for (int p = 0; p < k; ++p) d -= A[p * n + k] * A[p * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[k * n + k] = d;
        for (int j = k + 1; j < n; ++j) {
            double s = A[k * n + j];
            for (int p = 0; p < k; ++p) s -= A[p * n + k] * A[p * n + j];
            A[k * n + j] = s / d;
        }
    }
    return true;
}

void CholeskySolveUpper(const double* U, int n, double* b) {
    // U^T y = b
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int p = 0; p < i; ++p) s -= U[p * n + i] * b[p];
        b[i] = s / U[i * n + i];
    }
    // U x = y
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int p = i + 1; p < n; ++p) s -= U[i * n + p] * b[p];
        b[i] = s / U[i * n + i];
    }
}

bool InvertPSD3(const double* m, double* inv) {
    // InvertPSDMatrix<3>(assume_full_rank = true): m.selfadjointView<Upper>().llt().solve(Identity)
    double U[9];
    for (int i = 0; i < 9; ++i) U[i] = m[i];
    if (!CholeskyUpperInPlace(U, 3)) return false;
    for (int c = 0; c < 3; ++c) {
        double e[3] = {0, 0, 0};
        e[c] = 1.0;
        CholeskySolveUpper(U, 3, e);
        for (int r = 0; r < 3; ++r) inv[r * 3 + c] = e[r];
    }
    return true;
}

namespace {

using Clock = std::chrono::steady_clock;
inline double secs_since(Clock::time_point t0) {
    return std::chrono::duration<double>(Clock::now() - t0).count();
}

struct RowInfo {
    ResidualBlock* b;
    int nres;
    int np;
    int jac_sub[4];  // offset (relative to jac_off) of each param's local jacobian, -1 if constant
};

struct Program {
    std::vector<ParamBlock*> pblocks;  // e-blocks first
    int num_e = 0;
    int num_state = 0;  // ambient size
    int num_eff = 0;    // tangent size
    int num_f = 0;      // tangent size of the f part
    std::vector<RowInfo> rows;
    int num_residuals = 0;
    int jac_size = 0;
    std::vector<std::vector<int>> e_rows;  // per e-block: row indices
    std::vector<int> non_e_rows;
    double fixed_cost = 0.0;
};

struct Minimizer {
    const SolverOptions& opt;
    Problem& problem;
    Program& prog;
    SolverSummary& sum;

    std::vector<double> x, candidate_x, residuals, gradient, jac, scale, delta, step, model_res;
    std::vector<double> diagonal, lm_diagonal;
    double x_cost = std::numeric_limits<double>::max(), candidate_cost = 0, x_norm = -1, model_cost_change = 0;
    double radius, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int num_consecutive_invalid_steps = 0;
    IterationSummary it;
    Clock::time_point start;

    Minimizer(const SolverOptions& o, Problem& p, Program& g, SolverSummary& s) : opt(o), problem(p), prog(g), sum(s) {}

    void set_state(const std::vector<double>& s) {
        for (auto* pb : prog.pblocks) pb->state = s.data() + pb->state_off;
    }

    bool Plus(const std::vector<double>& xs, const std::vector<double>& d, std::vector<double>& out) {
        for (auto* pb : prog.pblocks) {
            if (!pb->Plus(xs.data() + pb->state_off, d.data() + pb->delta_off, out.data() + pb->state_off)) return false;
        }
        return true;
    }

    // Evaluator::Evaluate: cost, residuals (corrected), gradient (J^T r, unscaled), jacobian (local, corrected)
    bool Evaluate(const std::vector<double>& xs, double* cost, bool with_jac) {
        auto t0 = Clock::now();
        set_state(xs);
        const int nrows = (int)prog.rows.size();
        std::vector<double> costs(nrows, 0.0);
        int failed = 0;
        std::vector<double> tmp_res;
        double* res_out = residuals.data();
#pragma omp parallel for num_threads(opt.num_threads) schedule(static) reduction(+ : failed)
        for (int r = 0; r < nrows; ++r) {
            const RowInfo& ri = prog.rows[r];
            double* jl[4] = {nullptr, nullptr, nullptr, nullptr};
            if (with_jac) {
                for (int i = 0; i < ri.np; ++i)
                    if (ri.jac_sub[i] >= 0) jl[i] = jac.data() + ri.b->jac_off + ri.jac_sub[i];
            }
            double rloc[3];
            double* rp = with_jac ? res_out + ri.b->res_off : rloc;
            if (!problem.EvaluateBlock(*ri.b, true, &costs[r], rp, with_jac ? jl : nullptr)) failed++;
        }
        sum.time_eval += secs_since(t0);
        if (failed) return false;
        double c = 0.0;
        for (int r = 0; r < nrows; ++r) c += costs[r];  // fixed summation order
        *cost = c;
        if (with_jac) {
            std::fill(gradient.begin(), gradient.end(), 0.0);
            for (int r = 0; r < nrows; ++r) {
                const RowInfo& ri = prog.rows[r];
                const double* rr = residuals.data() + ri.b->res_off;
                for (int i = 0; i < ri.np; ++i) {
                    if (ri.jac_sub[i] < 0) continue;
                    const ParamBlock* pb = ri.b->params[i];
                    const int l = pb->lsize();
                    const double* J = jac.data() + ri.b->jac_off + ri.jac_sub[i];
                    for (int k = 0; k < ri.nres; ++k)
                        for (int c2 = 0; c2 < l; ++c2) gradient[pb->delta_off + c2] += J[k * l + c2] * rr[k];
                }
            }
        }
        return true;
    }

    void SquaredColumnNorm(std::vector<double>& out) {
        std::fill(out.begin(), out.end(), 0.0);
        for (const RowInfo& ri : prog.rows) {
            for (int i = 0; i < ri.np; ++i) {
                if (ri.jac_sub[i] < 0) continue;
                const ParamBlock* pb = ri.b->params[i];
                const int l = pb->lsize();
                const double* J = jac.data() + ri.b->jac_off + ri.jac_sub[i];
                for (int k = 0; k < ri.nres; ++k)
                    for (int c = 0; c < l; ++c) out[pb->delta_off + c] += J[k * l + c] * J[k * l + c];
            }
        }
    }
    void ScaleColumns(const std::vector<double>& s) {
        for (const RowInfo& ri : prog.rows) {
            for (int i = 0; i < ri.np; ++i) {
                if (ri.jac_sub[i] < 0) continue;
                const ParamBlock* pb = ri.b->params[i];
                const int l = pb->lsize();
                double* J = jac.data() + ri.b->jac_off + ri.jac_sub[i];
                for (int k = 0; k < ri.nres; ++k)
                    for (int c = 0; c < l; ++c) J[k * l + c] *= s[pb->delta_off + c];
            }
        }
    }

    bool EvaluateGradientAndJacobian() {
        if (!Evaluate(x, &x_cost, true)) {
            sum.message = "Residual and Jacobian evaluation failed.";
            sum.termination = FAILURE;
            return false;
        }
        it.cost = x_cost + sum.fixed_cost;
        if (opt.jacobi_scaling) {
            if (it.iteration == 0) {
                SquaredColumnNorm(scale);
                for (auto& s : scale) s = 1.0 / (1.0 + std::sqrt(s));
            }
            ScaleColumns(scale);
        }
        // |Plus(x, -gradient) - x|
        std::vector<double> neg(gradient.size()), proj(x.size());
        for (size_t i = 0; i < gradient.size(); ++i) neg[i] = -gradient[i];
        if (!Plus(x, neg, proj)) {
            sum.message = "projected_gradient_step = Plus(x, -gradient) failed.";
            sum.termination = FAILURE;
            return false;
        }
        double mx = 0.0, n2 = 0.0;
        for (auto* pb : prog.pblocks)
            for (int i = 0; i < pb->size; ++i) {
                const double d = x[pb->state_off + i] - proj[pb->state_off + i];
                mx = std::max(mx, std::fabs(d));
                n2 += d * d;
            }
        it.gradient_max_norm = mx;
        it.gradient_norm = std::sqrt(n2);
        return true;
    }

    static double norm_of(const std::vector<double>& v) {
        double s = 0.0;
        for (double e : v) s += e * e;
        return std::sqrt(s);
    }

    // SchurEliminator::Eliminate + dense Cholesky + BackSubstitute.  Solves min |J y - r|^2 + |D y|^2.
    // Returns false on LINEAR_SOLVER_FAILURE.
    bool LinearSolve(const std::vector<double>& D, std::vector<double>& y) {
        auto t0 = Clock::now();
        const int nf = prog.num_f, ne = prog.num_e;
        const int eoff = 0, foff = 3 * ne;
        std::vector<double> S((size_t)nf * nf, 0.0), rhs(nf, 0.0);
        const int nthreads = std::max(1, opt.num_linear_solver_threads);
        std::vector<std::vector<double>> Sth(nthreads), rth(nthreads);
        for (int t = 0; t < nthreads; ++t) {
            Sth[t].assign((size_t)nf * nf, 0.0);
            rth[t].assign(nf, 0.0);
        }
        std::vector<double> Vinv((size_t)ne * 9), ge((size_t)ne * 3);
        int bad = 0;
        auto accumulate_ff = [&](const RowInfo& ri, double* Sl, double* rl) {
            const double* rr = residuals.data() + ri.b->res_off;
            for (int a = 0; a < ri.np; ++a) {
                if (ri.jac_sub[a] < 0) continue;
                const ParamBlock* pa = ri.b->params[a];
                if (pa->e_index >= 0) continue;
                const int la = pa->lsize();
                const int oa = pa->delta_off - foff;
                const double* Ja = jac.data() + ri.b->jac_off + ri.jac_sub[a];
                for (int i = 0; i < la; ++i) {
                    double g = 0.0;
                    for (int k = 0; k < ri.nres; ++k) g += Ja[k * la + i] * rr[k];
                    rl[oa + i] += g;
                }
                for (int b = 0; b < ri.np; ++b) {
                    if (ri.jac_sub[b] < 0) continue;
                    const ParamBlock* pbk = ri.b->params[b];
                    if (pbk->e_index >= 0) continue;
                    const int lb = pbk->lsize();
                    const int ob = pbk->delta_off - foff;
                    const double* Jb = jac.data() + ri.b->jac_off + ri.jac_sub[b];
                    for (int i = 0; i < la; ++i)
                        for (int j = 0; j < lb; ++j) {
                            double s = 0.0;
                            for (int k = 0; k < ri.nres; ++k) s += Ja[k * la + i] * Jb[k * lb + j];
                            Sl[(size_t)(oa + i) * nf + ob + j] += s;
                        }
                }
            }
        };
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : bad)
        for (int e = 0; e < ne; ++e) {
            const int tid = omp_get_thread_num();
            double* Sl = Sth[tid].data();
            double* rl = rth[tid].data();
            double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            // W_a = sum F_a^T E for every f-block a in the chunk
            struct WB {
                int off, l;
                double w[10 * 3];
            };
            WB wb[72];  // f-blocks touched by one landmark: <= 3 per keyframe (pose, plane normal, plane distance), 20+ keyframes
            int nwb = 0;
            for (int r : prog.e_rows[e]) {
                const RowInfo& ri = prog.rows[r];
                const double* rr = residuals.data() + ri.b->res_off;
                int ei = -1;
                for (int a = 0; a < ri.np; ++a)
                    if (ri.jac_sub[a] >= 0 && ri.b->params[a]->e_index == e) ei = a;
                const double* E = jac.data() + ri.b->jac_off + ri.jac_sub[ei];
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) {
                        double s = 0.0;
                        for (int k = 0; k < ri.nres; ++k) s += E[k * 3 + i] * E[k * 3 + j];
                        V[i * 3 + j] += s;
                    }
                    double s = 0.0;
                    for (int k = 0; k < ri.nres; ++k) s += E[k * 3 + i] * rr[k];
                    g[i] += s;
                }
                accumulate_ff(ri, Sl, rl);
                for (int a = 0; a < ri.np; ++a) {
                    if (ri.jac_sub[a] < 0 || a == ei) continue;
                    const ParamBlock* pa = ri.b->params[a];
                    const int la = pa->lsize();
                    const int oa = pa->delta_off - foff;
                    int slot = -1;
                    for (int q = 0; q < nwb; ++q)
                        if (wb[q].off == oa) slot = q;
                    if (slot < 0) {
                        slot = nwb++;
                        wb[slot].off = oa;
                        wb[slot].l = la;
                        for (int q = 0; q < 30; ++q) wb[slot].w[q] = 0.0;
                    }
                    const double* Ja = jac.data() + ri.b->jac_off + ri.jac_sub[a];
                    for (int i = 0; i < la; ++i)
                        for (int j = 0; j < 3; ++j) {
                            double s = 0.0;
                            for (int k = 0; k < ri.nres; ++k) s += Ja[k * la + i] * E[k * 3 + j];
                            wb[slot].w[i * 3 + j] += s;
                        }
                }
            }
            for (int i = 0; i < 3; ++i) V[i * 3 + i] += D[eoff + 3 * e + i] * D[eoff + 3 * e + i];
            double* Vi = Vinv.data() + (size_t)e * 9;
            if (!InvertPSD3(V, Vi)) {
                bad++;
                continue;
            }
            for (int i = 0; i < 3; ++i) ge[(size_t)e * 3 + i] = g[i];
            // S -= W_a Vinv W_b^T ; rhs -= W_a Vinv g
            for (int a = 0; a < nwb; ++a) {
                double Y[30];
                for (int i = 0; i < wb[a].l; ++i)
                    for (int j = 0; j < 3; ++j)
                        Y[i * 3 + j] = wb[a].w[i * 3 + 0] * Vi[0 * 3 + j] + wb[a].w[i * 3 + 1] * Vi[1 * 3 + j] +
                                       wb[a].w[i * 3 + 2] * Vi[2 * 3 + j];
                for (int i = 0; i < wb[a].l; ++i)
                    rl[wb[a].off + i] -= Y[i * 3 + 0] * g[0] + Y[i * 3 + 1] * g[1] + Y[i * 3 + 2] * g[2];
                for (int b = 0; b < nwb; ++b)
                    for (int i = 0; i < wb[a].l; ++i)
                        for (int j = 0; j < wb[b].l; ++j)
                            Sl[(size_t)(wb[a].off + i) * nf + wb[b].off + j] -=
                                Y[i * 3 + 0] * wb[b].w[j * 3 + 0] + Y[i * 3 + 1] * wb[b].w[j * 3 + 1] +
                                Y[i * 3 + 2] * wb[b].w[j * 3 + 2];
            }
        }
        if (bad) {
            sum.time_schur += secs_since(t0);
            return false;
        }
        for (int r : prog.non_e_rows) accumulate_ff(prog.rows[r], Sth[0].data(), rth[0].data());
        for (int t = 0; t < nthreads; ++t) {
            for (size_t i = 0; i < S.size(); ++i) S[i] += Sth[t][i];
            for (int i = 0; i < nf; ++i) rhs[i] += rth[t][i];
        }
        for (int i = 0; i < nf; ++i) S[(size_t)i * nf + i] += D[foff + i] * D[foff + i];
        sum.time_schur += secs_since(t0);
        auto t1 = Clock::now();
        // SolveReducedLinearSystem: Eigen LLT on the upper triangle
        std::vector<double> yf(rhs);
        if (nf > 0) {
            if (!CholeskyUpperInPlace(S.data(), nf)) {
                sum.time_chol += secs_since(t1);
                return false;
            }
            CholeskySolveUpper(S.data(), nf, yf.data());
        }
        sum.time_chol += secs_since(t1);
        auto t2 = Clock::now();
        for (int i = 0; i < nf; ++i) y[foff + i] = yf[i];
        // BackSubstitute: y_e = Vinv (g_e - sum_a W_a^T y_a) recomputed from E, F like Ceres does
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int e = 0; e < ne; ++e) {
            double acc[3] = {ge[(size_t)e * 3 + 0], ge[(size_t)e * 3 + 1], ge[(size_t)e * 3 + 2]};
            for (int r : prog.e_rows[e]) {
                const RowInfo& ri = prog.rows[r];
                int ei = -1;
                for (int a = 0; a < ri.np; ++a)
                    if (ri.jac_sub[a] >= 0 && ri.b->params[a]->e_index == e) ei = a;
                const double* E = jac.data() + ri.b->jac_off + ri.jac_sub[ei];
                double fz[3] = {0, 0, 0};
                for (int a = 0; a < ri.np; ++a) {
                    if (ri.jac_sub[a] < 0 || a == ei) continue;
                    const ParamBlock* pa = ri.b->params[a];
                    const int la = pa->lsize();
                    const double* Ja = jac.data() + ri.b->jac_off + ri.jac_sub[a];
                    for (int k = 0; k < ri.nres; ++k)
                        for (int i = 0; i < la; ++i) fz[k] += Ja[k * la + i] * yf[pa->delta_off - foff + i];
                }
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < ri.nres; ++k) acc[i] -= E[k * 3 + i] * fz[k];
            }
            const double* Vi = Vinv.data() + (size_t)e * 9;
            for (int i = 0; i < 3; ++i) y[eoff + 3 * e + i] = Vi[i * 3 + 0] * acc[0] + Vi[i * 3 + 1] * acc[1] + Vi[i * 3 + 2] * acc[2];
        }
        sum.time_schur += secs_since(t2);
        for (double v : y)
            if (!std::isfinite(v)) return false;
        return true;
    }

    // LevenbergMarquardtStrategy::ComputeStep + TrustRegionMinimizer::ComputeTrustRegionStep
    void ComputeTrustRegionStep() {
        if (!reuse_diagonal) {
            SquaredColumnNorm(diagonal);
            for (auto& d : diagonal) d = std::min(std::max(d, opt.min_lm_diagonal), opt.max_lm_diagonal);
        }
        for (size_t i = 0; i < diagonal.size(); ++i) lm_diagonal[i] = std::sqrt(diagonal[i] / radius);
        it.step_is_valid = false;
        bool ok = LinearSolve(lm_diagonal, step);
        reuse_diagonal = true;
        if (opt.probe && !opt.probe->filled) {
            StepProbe& pr = *opt.probe;
            pr.filled = true;
            pr.num_residuals = prog.num_residuals;
            pr.num_eff = prog.num_eff;
            pr.num_e = prog.num_e;
            pr.J.assign((size_t)prog.num_residuals * prog.num_eff, 0.0);
            for (const RowInfo& ri : prog.rows)
                for (int i = 0; i < ri.np; ++i) {
                    if (ri.jac_sub[i] < 0) continue;
                    const ParamBlock* pb = ri.b->params[i];
                    const int l = pb->lsize();
                    const double* J = jac.data() + ri.b->jac_off + ri.jac_sub[i];
                    for (int k = 0; k < ri.nres; ++k)
                        for (int c = 0; c < l; ++c) pr.J[(size_t)(ri.b->res_off + k) * prog.num_eff + pb->delta_off + c] = J[k * l + c];
                }
            pr.r = residuals;
            pr.D = lm_diagonal;
            pr.y = ok ? step : std::vector<double>();
        }
        if (!ok) return;
        for (auto& s : step) s = -s;
        // model_cost_change = -(J step)'(f + J step / 2)
        double mcc = 0.0;
        for (const RowInfo& ri : prog.rows) {
            double m[3] = {0, 0, 0};
            for (int i = 0; i < ri.np; ++i) {
                if (ri.jac_sub[i] < 0) continue;
                const ParamBlock* pb = ri.b->params[i];
                const int l = pb->lsize();
                const double* J = jac.data() + ri.b->jac_off + ri.jac_sub[i];
                for (int k = 0; k < ri.nres; ++k)
                    for (int c = 0; c < l; ++c) m[k] += J[k * l + c] * step[pb->delta_off + c];
            }
            const double* rr = residuals.data() + ri.b->res_off;
            for (int k = 0; k < ri.nres; ++k) mcc -= m[k] * (rr[k] + m[k] / 2.0);
        }
        model_cost_change = mcc;
        it.step_is_valid = (model_cost_change > 0.0);
        if (it.step_is_valid) {
            for (size_t i = 0; i < step.size(); ++i) delta[i] = step[i] * scale[i];
            num_consecutive_invalid_steps = 0;
        }
    }

    bool Finalize() {  // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (it.step_is_successful) {
            ++sum.num_successful_steps;
        } else {
            ++sum.num_unsuccessful_steps;
        }
        it.trust_region_radius = radius;
        sum.iterations.push_back(it);
        if (opt.max_solver_time_in_seconds > 0 && secs_since(start) >= opt.max_solver_time_in_seconds) {
            sum.message = "Maximum solver time reached.";
            sum.termination = NO_CONVERGENCE;
            return false;
        }
        if (it.iteration >= opt.max_num_iterations) {
            sum.message = "Maximum number of iterations reached.";
            sum.termination = NO_CONVERGENCE;
            return false;
        }
        if (it.step_is_successful && it.gradient_max_norm <= opt.gradient_tolerance) {
            sum.message = "Gradient tolerance reached.";
            sum.termination = CONVERGENCE;
            return false;
        }
        if (it.trust_region_radius <= opt.min_trust_region_radius) {
            sum.message = "Minimum trust region radius reached.";
            sum.termination = CONVERGENCE;
            return false;
        }
        return true;
    }

    void Run() {
        start = Clock::now();
        const int ns = prog.num_state, ne = prog.num_eff;
        x.assign(ns, 0.0);
        candidate_x.assign(ns, 0.0);
        for (auto* pb : prog.pblocks)
            for (int i = 0; i < pb->size; ++i) x[pb->state_off + i] = pb->user[i];
        residuals.assign(prog.num_residuals, 0.0);
        gradient.assign(ne, 0.0);
        jac.assign(prog.jac_size, 0.0);
        scale.assign(ne, 1.0);
        delta.assign(ne, 0.0);
        step.assign(ne, 0.0);
        diagonal.assign(ne, 0.0);
        lm_diagonal.assign(ne, 0.0);
        radius = opt.initial_trust_region_radius;
        sum.termination = NO_CONVERGENCE;

        // IterationZero
        it = IterationSummary();
        it.iteration = 0;
        x_norm = norm_of(x);
        bool ok = EvaluateGradientAndJacobian();
        if (ok) {
            sum.initial_cost = x_cost + sum.fixed_cost;
            it.step_is_valid = true;
            it.step_is_successful = true;
            while (Finalize()) {
                const int next = sum.iterations.back().iteration + 1;
                it = IterationSummary();
                it.iteration = next;
                ComputeTrustRegionStep();
                if (!it.step_is_valid) {  // HandleInvalidStep
                    ++num_consecutive_invalid_steps;
                    if (num_consecutive_invalid_steps >= opt.max_num_consecutive_invalid_steps) {
                        sum.message = "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps";
                        sum.termination = FAILURE;
                        break;
                    }
                    radius = radius / decrease_factor;  // StepIsInvalid
                    decrease_factor *= 2.0;
                    reuse_diagonal = false;
                    it.cost = x_cost + sum.fixed_cost;
                    it.gradient_max_norm = sum.iterations.back().gradient_max_norm;
                    it.gradient_norm = sum.iterations.back().gradient_norm;
                    continue;
                }
                // ComputeCandidatePointAndEvaluateCost
                if (!Plus(x, delta, candidate_x)) {
                    candidate_cost = std::numeric_limits<double>::max();
                } else if (!Evaluate(candidate_x, &candidate_cost, false)) {
                    candidate_cost = std::numeric_limits<double>::max();
                }
                // ParameterToleranceReached
                {
                    double s = 0.0;
                    for (int i = 0; i < ns; ++i) s += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
                    it.step_norm = std::sqrt(s);
                    const double tol = opt.parameter_tolerance * (x_norm + opt.parameter_tolerance);
                    if (it.step_norm <= tol) {
                        sum.message = "Parameter tolerance reached.";
                        sum.termination = CONVERGENCE;
                        break;
                    }
                }
                // FunctionToleranceReached
                it.cost_change = x_cost - candidate_cost;
                if (std::fabs(it.cost_change) <= opt.function_tolerance * x_cost) {
                    sum.message = "Function tolerance reached.";
                    sum.termination = CONVERGENCE;
                    break;
                }
                // IsStepSuccessful
                it.relative_decrease = (x_cost - candidate_cost) / model_cost_change;
                if (it.relative_decrease > opt.min_relative_decrease) {  // HandleSuccessfulStep
                    x = candidate_x;
                    x_norm = norm_of(x);
                    if (!EvaluateGradientAndJacobian()) break;
                    it.step_is_successful = true;
                    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
                    radius = std::min(opt.max_trust_region_radius, radius);
                    decrease_factor = 2.0;
                    reuse_diagonal = false;
                } else {  // HandleUnsuccessfulStep
                    it.step_is_successful = false;
                    radius = radius / decrease_factor;
                    decrease_factor *= 2.0;
                    reuse_diagonal = true;
                    it.cost = candidate_cost + sum.fixed_cost;
                    it.gradient_max_norm = sum.iterations.back().gradient_max_norm;
                    it.gradient_norm = sum.iterations.back().gradient_norm;
                }
            }
        }
        if (sum.termination != FAILURE) {
            for (auto* pb : prog.pblocks)
                for (int i = 0; i < pb->size; ++i) pb->user[i] = x[pb->state_off + i];
            sum.final_cost = x_cost + sum.fixed_cost;
        } else {
            sum.final_cost = sum.initial_cost;
        }
        for (auto* pb : prog.pblocks) pb->state = nullptr;
        sum.time_total = secs_since(start);
    }
};

}  // namespace

void Solve(const SolverOptions& options, Problem* problem, SolverSummary* summary) {
    *summary = SolverSummary();
    Program prog;
    // Reduced program (program.cc RemoveFixedBlocks): drop residual blocks whose parameters are all constant
    // (their cost is the fixed cost) and parameter blocks that are constant or unused.
    for (auto& pb : problem->params) {
        pb->state_off = pb->delta_off = pb->e_index = -1;
        pb->state = nullptr;
    }
    std::vector<ResidualBlock*> active;
    double fixed_cost = 0.0;
    for (auto& b : problem->blocks) {
        if (b->removed) continue;
        bool any = false;
        for (auto* p : b->params)
            if (!p->constant) any = true;
        if (!any) {
            double c, r[3];
            if (!problem->EvaluateBlock(*b, true, &c, r, nullptr)) {
                summary->termination = FAILURE;
                summary->message = "Evaluation of the residual of a fixed block failed.";
                return;
            }
            fixed_cost += c;
            continue;
        }
        active.push_back(b.get());
    }
    summary->fixed_cost = fixed_cost;
    // parameter blocks in order of first use; e-blocks (landmarks) first
    std::vector<ParamBlock*> eb, fb;
    for (auto* b : active)
        for (auto* p : b->params) {
            if (p->constant || p->state_off != -1) continue;
            p->state_off = 0;  // mark
            (p->is_landmark ? eb : fb).push_back(p);
        }
    if (eb.empty() && fb.empty()) {
        summary->termination = CONVERGENCE;
        summary->message = "Function tolerance reached. No non-constant parameter blocks found.";
        summary->initial_cost = summary->final_cost = fixed_cost;
        return;
    }
    int so = 0, doff = 0;
    for (size_t i = 0; i < eb.size(); ++i) {
        eb[i]->e_index = (int)i;
        eb[i]->state_off = so;
        eb[i]->delta_off = doff;
        so += eb[i]->size;
        doff += eb[i]->lsize();
        prog.pblocks.push_back(eb[i]);
    }
    prog.num_e = (int)eb.size();
    const int fstart = doff;
    for (auto* p : fb) {
        p->state_off = so;
        p->delta_off = doff;
        so += p->size;
        doff += p->lsize();
        prog.pblocks.push_back(p);
    }
    prog.num_state = so;
    prog.num_eff = doff;
    prog.num_f = doff - fstart;
    prog.e_rows.resize(prog.num_e);
    int ro = 0, jo = 0;
    for (auto* b : active) {
        RowInfo ri;
        ri.b = b;
        ri.nres = b->cost->nres;
        ri.np = (int)b->params.size();
        b->res_off = ro;
        b->jac_off = jo;
        int sub = 0, e = -1;
        for (int i = 0; i < 4; ++i) ri.jac_sub[i] = -1;
        for (int i = 0; i < ri.np; ++i) {
            ParamBlock* p = b->params[i];
            if (p->constant) continue;
            ri.jac_sub[i] = sub;
            sub += ri.nres * p->lsize();
            if (p->e_index >= 0) e = p->e_index;
        }
        ro += ri.nres;
        jo += sub;
        if (e >= 0)
            prog.e_rows[e].push_back((int)prog.rows.size());
        else
            prog.non_e_rows.push_back((int)prog.rows.size());
        prog.rows.push_back(ri);
    }
    prog.num_residuals = ro;
    prog.jac_size = jo;
    summary->num_e_blocks = prog.num_e;
    summary->num_f_params = prog.num_f;
    summary->num_residuals = ro;
    Minimizer m(options, *problem, prog, *summary);
    m.Run();
}

}  // namespace kba_oracle
