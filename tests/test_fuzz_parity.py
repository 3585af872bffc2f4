"""Randomised parity sweep against the oracle (scripts/gpu_fuzz.py promoted to tests).

CPU tier: the emulated pipeline (same statements as the gfx950 kernels) on a slice of the sweep that contains the known
plateau window (seed 77, window 21).  GPU tier: liblimo_hip.so through the C-ABI on the full sweeps - seed 123 x 290
windows, seed 77 x 60 and seed 2026 x 240 windows (LIMO_FUZZ_SCALE=0.1 runs a tenth of them for a quick check) - single solves against
the oracle, then one mixed batch that must reproduce the single solves bit for bit.
The parity rule is fuzz_common.check_parity.
"""
import os

import numpy as np
import pytest

import fuzz_common as fc
from limo_amd import default_options


def _oracle_solver(oracle, threads):
    def solve(w, o):
        rep, _ = oracle.solve(w, o, num_threads=threads)
        return rep

    return solve


def _emu_solver(emu):
    def solve(w, o):
        return emu.solve_batch([w], o)[0]

    return solve


def test_emulated_pipeline_fuzz_vs_oracle(oracle, emu):
    o = default_options()
    so, se = _oracle_solver(oracle, 8), _emu_solver(emu)
    n_plateau = 0
    for i, (kw, w) in enumerate(fc.random_windows(24, 77)):
        we, wo = w.copy(), w.copy()
        re_ = se(we, o)
        te = emu.last_trimmed(0)
        ro = so(wo, o)
        to = oracle.last_trimmed()
        ok, detail, clause = fc.check_parity(w, re_, we, te, se, ro, wo, to, so)
        assert ok, "window %d %r: %s" % (i, kw, detail)
        assert clause != fc.BY_CAP
        n_plateau += clause == fc.BY_SPREAD
    assert n_plateau <= 1  # at most window 21 (the documented plateau case) needs the 1-ulp-spread rule


def test_final_cost_of_outlier_windows_is_not_determined_to_1e4(oracle):
    """The justification of the 1-ulp-spread rule, pinned: on the two windows where implementations disagree on the
    final cost the ORACLE's own cost moves by more than 1e-4 when one input coordinate moves by 1 ulp - while its poses
    stay within the 1e-4 bar."""
    so = _oracle_solver(oracle, 8)
    for seed, idx in ((77, 21), (123, 115)):
        kw, w = fc.random_windows(idx + 1, seed)[idx]
        sc, sp = fc.ulp_spread(w, so)
        assert sc > fc.TOL and sp <= fc.TOL, (seed, idx, sc, sp)


# Windows of a third sweep (scripts/gpu_fuzz.py 400 2026) whose RESULT is not determined to 1e-4 by their input: the oracle's own
# poses move by 1e-2 (11, 225: twelve keyframes on 60 landmarks, mono, no depth, no ground plane) or its cost by 1e-2 (89, 96:
# 15 % gross outliers) when one input coordinate moves by 1 ulp.
WEAK_2026 = {
    11: dict(n_kf=12, n_lm=60, depth_prob=0.02, ground_frac=0.05, outlier_frac=0.15, stereo_baseline=0.0, with_ground_plane=False),
    89: dict(n_kf=11, n_lm=3000, depth_prob=0.2, ground_frac=0.05, outlier_frac=0.15, stereo_baseline=0.0, with_ground_plane=True),
    96: dict(n_kf=11, n_lm=1500, depth_prob=0.9, ground_frac=0.05, outlier_frac=0.15, stereo_baseline=0.54, with_ground_plane=True),
    225: dict(n_kf=12, n_lm=60, depth_prob=0.0, ground_frac=0.05, outlier_frac=0.05, stereo_baseline=0.0, with_ground_plane=False),
}


@pytest.mark.parametrize("idx", sorted(WEAK_2026))
def test_weakly_determined_windows_stay_inside_the_oracles_own_spread(oracle, emu, idx):
    """The parity rule on the windows a 400-window sweep with a fresh seed flagged: the strict bar cannot hold (the oracle
    does not reproduce ITSELF to 1e-4 on them), the emulated pipeline has to stay within 3x the oracle's own 1-ulp spread."""
    from limo_amd import synth

    w = synth.make_window(10000 + idx, **WEAK_2026[idx])
    o = default_options()
    so, se = _oracle_solver(oracle, 8), _emu_solver(emu)
    we, wo = w.copy(), w.copy()
    re_ = se(we, o)
    te = emu.last_trimmed(0)
    ro = so(wo, o)
    to = oracle.last_trimmed()
    s = fc.input_sensitivity(w, so)
    assert s["cost"] > fc.TOL or s["pose"] > fc.TOL or len(s["terminations"]) > 1, s  # (not determined by its input)
    ok, detail, clause = fc.check_parity(w, re_, we, te, se, ro, wo, to, so)
    assert ok and clause != fc.BY_CAP, "window %d: %s" % (idx, detail)


def test_window_at_the_iteration_cap_agrees_on_everything_but_the_count(oracle, emu):
    """Seed 5151 window 116 (a fresh sweep of round 5): eight keyframes, 3000 landmarks, mono, no depth.  The oracle needs 93 LM
    iterations, the emulated pipeline 94, the gfx950 kernels 99 (round 4) / 100 = max_num_iterations (round 5: NO_CONVERGENCE) -
    with the same trimmed set, cost and poses inside 1e-4.  Pinned here: the solve crawls (half of its steps are rejected), both CPU
    implementations end inside the last tenth of the budget, and they agree with each other on the strict rule."""
    kw, w = fc.random_windows(117, 5151)[116]
    o = default_options()
    so, se = _oracle_solver(oracle, 8), _emu_solver(emu)
    we, wo = w.copy(), w.copy()
    re_, ro = se(we, o), so(wo, o)
    assert min(re_["iterations_total"], ro["iterations_total"]) >= 90
    assert ro["successful_steps"] <= 0.6 * ro["iterations_total"]
    ok, detail, _ = fc.check_parity(w, re_, we, emu.last_trimmed(0), se, ro, wo, oracle.last_trimmed(), so)
    assert ok, detail


# LIMO_FUZZ_EXTRA="seed:n[,seed:n]" adds sweeps with fresh seeds (one-off evidence after kernel changes; profiles/r04_fuzz_*.log)
# (seed 5151 joined the tier in round 6: the sweep that found the iteration-cap window on the round-5 kernels)
_SWEEPS = [(123, 290), (77, 60), (2026, 240), (5151, 320)] + [tuple(int(x) for x in e.split(":")) for e in os.environ.get("LIMO_FUZZ_EXTRA", "").split(",") if e]
POSE_WATCH = 5e-5  # half the 1e-4 bar: a sweep whose worst strictly-judged pose error comes this close fails, so that the margin is watched


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", _SWEEPS)
def test_gpu_fuzz_vs_oracle(ctx, oracle, seed, n):
    from limo_amd import ba

    scale = float(os.environ.get("LIMO_FUZZ_SCALE", "1"))
    n = max(4, int(round(n * scale))) if scale < 1 else n
    o = default_options()
    so = _oracle_solver(oracle, 8)

    def sg(w, opts):
        return ctx.solve(w, opts)

    cases = fc.random_windows(n, seed)
    singles, n_plateau, n_cap, n_ill, n_watch, worst_c, worst_p, worst_at = [], 0, 0, 0, 0, 0.0, 0.0, -1
    for i, (kw, w) in enumerate(cases):
        wg, wo = w.copy(), w.copy()
        b = ba.Batch(ctx, [wg])  # a batch of one: same kernels as limo_ba_solve, and the trimmed set can be read back
        b.solve(o)
        rg = b.download()[0]
        tg = b.trimmed(0)
        b.close()
        ro = so(wo, o)
        to = oracle.last_trimmed()
        ok, detail, clause = fc.check_parity(w, rg, wg, tg, sg, ro, wo, to, so)
        assert ok, "seed %d window %d %r: %s" % (seed, i, kw, detail)
        n_plateau += clause == fc.BY_SPREAD
        n_cap += clause == fc.BY_CAP
        if clause != fc.STRICT:
            print("fuzz seed %d window %d accepted by clause %d: %s" % (seed, i, clause, detail))
        posed = fc.well_posed(w)
        n_ill += not posed
        if clause != fc.BY_SPREAD and posed:  # (windows whose result the input does not determine are judged by the spread rule)
            worst_c = max(worst_c, fc.rel_cost_err(rg, ro))
            ep = fc.rel_pose_err(wg.kf_pose, wo.kf_pose)
            if ep > POSE_WATCH:
                # inside the bar but close to it: is that the kernels' arithmetic, or does the INPUT not determine the poses any
                # better?  (seed 123 window 53 - ten keyframes, stereo, 2 % depth: the oracle's own poses move by 6.5e-5 under a
                # 1-ulp change of one input coordinate, and the IEEE emulation differs from it by exactly as much)
                s_own = fc.input_sensitivity(w, so, n=16)
                print("fuzz seed %d window %d: rel pose %.2e above the watch level; the oracle's own 1-ulp spread is %.2e" % (seed, i, ep, s_own["pose"]))
                assert ep <= 3.0 * s_own["pose"], "seed %d window %d: rel pose %.2e is within 2x of the 1e-4 bar and NOT explained by the oracle's own spread (%.2e)" % (seed, i, ep, s_own["pose"])
                n_watch += 1
            elif ep > worst_p:
                worst_p, worst_at = ep, i
        singles.append(wg)
    # windows whose result is not determined to 1e-4 by their input are rare: 1 in 290 / 1 in 60 / 4 in 240 at full size
    assert n_plateau <= max(1, n // 40)
    # the iteration-cap clause (a termination type that differs from the reference's: fuzz_common.BY_CAP) has its own, tighter count
    assert n_cap <= 1, n_cap
    # the margin to the 1e-4 bar is watched, not discovered: a strictly-judged window above POSE_WATCH must be explained by the
    # oracle's own spread (checked above, window by window), and such windows stay rare
    assert worst_p <= POSE_WATCH and n_watch <= max(1, n // 100), (worst_p, n_watch)
    # windows that only get the weak checks (a keyframe with < 8 observations: 9 of 290 / 1 of 60 at full size) stay few:
    # the strict rule (sets, termination, pose and cost to 1e-4) covers >= 95 % of the sweep
    assert n_ill <= max(1, n // 20) or scale < 1, n_ill  # (a property of the sample: only meaningful at full size)
    # kernel variants are chosen per window: a mixed batch reproduces every single solve bit for bit
    b = ba.Batch(ctx, [w.copy() for _, w in cases])
    b.solve(o)
    b.download()
    for i, (ws, wb) in enumerate(zip(singles, b.windows)):
        assert np.array_equal(ws.kf_pose, wb.kf_pose) and np.array_equal(ws.lm_pos, wb.lm_pos), "batch != single, window %d" % i
    b.close()
    print("fuzz seed %d: %d windows, %d plateau cases, %d at the iteration cap, %d ill-posed (weak checks), %d above the pose watch level (explained by the oracle's own spread), worst rel cost %.2e, worst rel pose of the others %.2e (window %d)"
          % (seed, n, n_plateau, n_cap, n_ill, n_watch, worst_c, worst_p, worst_at))
