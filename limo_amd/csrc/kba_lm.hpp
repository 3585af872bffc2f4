// kba_lm.hpp — per-window Levenberg–Marquardt control, one lane per window (device) or a plain loop (emulator).
//
// Replaces the control flow ceres::Solve runs for the reference (call sites
// robust_optimization/src/robust_solving.cpp:169,174,239): Ceres 1.13 TrustRegionMinimizer::Minimize with
// LevenbergMarquardtStrategy — IterationZero, ComputeTrustRegionStep validity, parameter / function tolerance,
// step acceptance (min_relative_decrease), radius update radius/max(1/3, 1-(2 rho-1)^3), step rejection
// radius/decrease_factor (factor doubling), invalid-step handling, iteration / gradient / min-radius stops.
// The arithmetic-heavy parts (cost, model cost change, norms) arrive as reductions in WinRed.
#pragma once
#include <float.h>
#include <math.h>

#include "kba_layout.hpp"
#include "kba_math.hpp"

namespace kba {

// Start a ceres-style solve for window w (if selected by `selected`).
KBA_HD void lm_solve_init(WinState& s, bool selected, int max_iter, const SolveConsts& c) {
    s.in_phase = selected ? 1 : 0;
    s.active = selected ? 1 : 0;
    s.need_lin = selected ? 1 : 0;
    s.first = 1;
    s.accept = 0;
    s.iter = 0;
    s.max_iter = max_iter;
    s.term = selected ? -1 : s.term;
    s.invalid_run = 0;
    s.compute_scale = 1;
    s.redamp = 0;
    s.n_success = 0;
    s.n_unsuccess = 0;
    s.radius = c.initial_radius;
    s.decrease_factor = 2.0;
    if (selected) s.acc_solves += 1;
}

KBA_HD void lm_terminate(WinState& s, int term) {
    s.term = term;
    s.active = 0;
    s.solve_final_cost = (term == LIMO_FAILURE && s.first) ? s.solve_initial_cost : s.x_cost + s.fixed_cost;
    s.acc_iters += s.iter;
    s.acc_success += s.n_success > 0 ? s.n_success - 1 : 0;
    s.last_iters = s.iter;
}

// FinalizeIterationAndCheckIfMinimizerCanContinue for an unsuccessful / invalid iteration
KBA_HD void lm_finalize_unsuccessful(WinState& s, const SolveConsts& c) {
    s.n_unsuccess += 1;
    if (s.iter >= s.max_iter) {
        lm_terminate(s, LIMO_NO_CONVERGENCE);
    } else if (s.radius <= c.min_radius) {
        lm_terminate(s, LIMO_CONVERGENCE);
    }
}

// Streaming solve: advance window w in the solveTrimmed schedule at the start of a round (the lock-step form of the
// same schedule is kba_pack.cpp:run_schedule).  Returns 0 = nothing to do this round (cannot happen for a window in a
// slot), 1 = the window takes part in this round - it iterates, or (phase == PH_TRIM) it is trimmed during this round
// and re-armed by sched_after_trim for the next one, 2 = finished: the slot is free.
KBA_HD int sched_advance(WinState& s, const WinDesc& wd, const SolveConsts& c) {
    if (s.phase == PH_IDLE) {  // just moved into a slot
        s.trim_round = 0;
        if (wd.do_trim && c.num_trim_rounds > 0) {
            s.phase = PH_TRIM_SOLVE;
            lm_solve_init(s, true, c.trim_iters, c);
        } else {
            s.phase = PH_FINAL;
            lm_solve_init(s, true, c.max_iters, c);
        }
        return 1;
    }
    if (s.active || s.phase == PH_TRIM) return 1;
    if (s.phase == PH_TRIM_SOLVE) {
        if (s.solve_initial_cost - s.solve_final_cost <= 0.0) {  // robust_solving.cpp:172-181
            s.phase = PH_RETRY;
            lm_solve_init(s, true, 3 * c.trim_iters, c);
        } else {
            s.phase = PH_TRIM;
        }
        return 1;
    }
    if (s.phase == PH_RETRY) {
        s.phase = PH_TRIM;
        return 1;
    }
    if (s.phase == PH_FINAL) {
        s.phase = PH_DONE;
        return 2;
    }
    return 0;
}

// After the trimming kernels of a round have removed the outliers of window w: start its next solve.
KBA_HD void sched_after_trim(WinState& s, const SolveConsts& c) {
    s.trim_round += 1;
    if (s.trim_round < c.num_trim_rounds) {
        s.phase = PH_TRIM_SOLVE;
        lm_solve_init(s, true, c.trim_iters, c);
    } else {
        s.phase = PH_FINAL;
        lm_solve_init(s, true, c.max_iters, c);
    }
}

// After (re)linearisation at the current point: IterationZero or the tail of HandleSuccessfulStep, then Finalize.
KBA_HD void lm_decide_lin(WinState& s, const WinRed& r, double fixed_cost, const SolveConsts& c) {
    if (!s.active || !s.need_lin) return;
    s.need_lin = 0;
    s.accept = 0;
    s.compute_scale = 0;
    if (r.lin_fail) {  // "Residual and Jacobian evaluation failed."
        if (s.first) {
            s.fixed_cost = fixed_cost;
            s.solve_initial_cost = -1.0;
            if (s.acc_solves == 1) s.first_initial_cost = -1.0;
        }
        lm_terminate(s, LIMO_FAILURE);
        s.first = 0;
        return;
    }
    // the cost at an accepted candidate is already known (HandleSuccessfulStep: x_cost = candidate_cost); only the
    // first linearisation of a solve evaluates it (kernels skip the cost value otherwise)
    s.x_cost = s.first ? r.lin_cost : s.cost_pending;
    s.gmax = r.gmax;
    s.acc_lin += 1;
    if (s.first) {
        s.first = 0;
        s.fixed_cost = fixed_cost;
        s.x_norm = sqrt(r.xnorm2);
        s.solve_initial_cost = s.x_cost + s.fixed_cost;
        if (s.acc_solves == 1) s.first_initial_cost = s.solve_initial_cost;
        s.n_success = 1;
    } else {
        s.x_norm = s.xnorm_pending;  // (radius and decrease factor of the successful step: set in lm_decide_step already)
        s.n_success += 1;
    }
    if (s.iter >= s.max_iter) {
        lm_terminate(s, LIMO_NO_CONVERGENCE);
    } else if (s.gmax <= c.gradient_tolerance) {
        lm_terminate(s, LIMO_CONVERGENCE);
    } else if (s.radius <= c.min_radius) {
        lm_terminate(s, LIMO_CONVERGENCE);
    }
}

// After a trust-region step was computed and the candidate point evaluated.
KBA_HD void lm_decide_step(WinState& s, const WinRed& r, const SolveConsts& c) {
    if (!s.active) return;
    s.iter += 1;
    const bool finite_ok = isfinite(r.mcc) && isfinite(r.step2) && isfinite(r.cand2);
    const bool valid = !r.chol_fail && finite_ok && (r.mcc > 0.0);
    if (!valid) {  // HandleInvalidStep
        s.invalid_run += 1;
        if (s.invalid_run >= c.max_invalid) {
            lm_terminate(s, LIMO_FAILURE);
            return;
        }
        s.radius = s.radius / s.decrease_factor;  // StepIsInvalid
        s.decrease_factor *= 2.0;
        s.redamp = 1;
        lm_finalize_unsuccessful(s, c);
        return;
    }
    s.invalid_run = 0;
    const double cand_cost = r.cand_fail ? DBL_MAX : r.cand_cost;
    const double step_norm = sqrt(r.step2);
    if (step_norm <= c.parameter_tolerance * (s.x_norm + c.parameter_tolerance)) {
        lm_terminate(s, LIMO_CONVERGENCE);
        return;
    }
    const double cost_change = s.x_cost - cand_cost;
    if (fabs(cost_change) <= c.function_tolerance * s.x_cost) {
        lm_terminate(s, LIMO_CONVERGENCE);
        return;
    }
    const double rho = cost_change / r.mcc;
    if (rho > c.min_relative_decrease) {  // HandleSuccessfulStep (completed in lm_decide_lin)
        s.accept = 1;
        s.need_lin = 1;
        s.redamp = 0;
        // the trust region of the next step (HandleSuccessfulStep -> LevenbergMarquardtStrategy::StepAccepted): fixed here
        // rather than after the relinearisation, so that the landmark pass of the relinearisation can damp with it
        {
            const double q = 2.0 * rho - 1.0;
            s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
            s.radius = fmin(c.max_radius, s.radius);
            s.decrease_factor = 2.0;
        }
        s.rho_pending = rho;
        s.xnorm_pending = sqrt(r.cand2);
        s.cost_pending = cand_cost;
    } else {  // HandleUnsuccessfulStep
        s.radius = s.radius / s.decrease_factor;
        s.decrease_factor *= 2.0;
        s.redamp = 1;
        lm_finalize_unsuccessful(s, c);
    }
}

}  // namespace kba
