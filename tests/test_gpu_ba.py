"""GPU parity tests of the keyframe-BA path: every call goes through the C-ABI of liblimo_hip.so.

Bar (BASELINE.json north_star): relative error <= 1e-4 on every free keyframe translation and on the final
cost versus the oracle on identical input bytes; per-entry Jacobian parity is held to 1e-9 relative.
"""
import numpy as np
import pytest

from limo_amd import ba, default_options, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star tolerance on pose translation and final cost


def rel_pose_err(a, b):
    return np.abs(a[:, 4:] - b[:, 4:]).max() / max(1e-12, np.abs(b[:, 4:]).max())


def check_solve_parity(ctx, oracle, w, opts, tol=TOL):
    wg, wo = w.copy(), w.copy()
    rg = ctx.solve(wg, opts)
    ro, _ = oracle.solve(wo, opts)
    assert rg["n_depth_blocks"] == ro["n_depth_blocks"] and rg["n_gp_blocks"] == ro["n_gp_blocks"]
    assert rg["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"]
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-9 * abs(ro["initial_cost"])
    assert abs(rg["final_cost"] - ro["final_cost"]) <= tol * abs(ro["final_cost"])
    assert rel_pose_err(wg.kf_pose, wo.kf_pose) <= tol
    # keyframe 0 is Pose-fixed: must come back bit-identical
    assert np.array_equal(wg.kf_pose[0], w.kf_pose[0])
    return rg, ro, wg, wo


@pytest.mark.parametrize("apply_loss", [False, True])
def test_evaluate_matches_oracle(ctx, oracle, apply_loss):
    w = synth.config_c2()
    o = default_options()
    c0, r0, jp0, jl0, v0 = oracle.evaluate(w, o, apply_loss)
    c1, r1, jp1, jl1, v1 = ctx.evaluate(w, o, apply_loss)
    assert np.array_equal(v0, v1)
    assert abs(c0 - c1) <= 1e-12 * abs(c0)
    assert np.abs(r0 - r1).max() <= 1e-9 * max(1.0, np.abs(r0).max())
    assert np.abs(jp0 - jp1).max() <= 1e-9 * np.abs(jp0).max()
    assert np.abs(jl0 - jl1).max() <= 1e-9 * np.abs(jl0).max()


@pytest.mark.parametrize("n_lm,depth_prob,n_kf,stereo", [
    (1, 1.0, 3, 0.0), (8, 0.0, 2, 0.0), (64, 0.5, 3, 0.0), (65, 1.0, 3, 0.0), (70, 0.45, 3, 0.54), (129, 0.02, 4, 0.0),
    (700, 0.9, 5, 0.0), (1300, 0.45, 3, 0.0), (3000, 0.2, 2, 0.54)])
def test_evaluate_layout_edge_cases(ctx, oracle, n_lm, depth_prob, n_kf, stereo):
    """The materialised pass writes the rows that exist through wave-sized chunks of 64-aligned observation ranges, compact depth
    planes and a per-workgroup staging buffer (kba_kernels.hip:k_evaluate): windows whose views end ON a chunk boundary, one
    observation past it, inside the first chunk; views of more than 1024 observations (two observation blocks); no depth at all, a
    depth on every observation; two cameras per keyframe.  Every entry against the oracle's dual numbers, <= 1e-9, and the flags and
    the total cost exactly as the oracle has them."""
    w = synth.make_window(4000 + n_lm, n_kf=n_kf, n_lm=n_lm, depth_prob=depth_prob, stereo_baseline=stereo)
    o = default_options()
    for apply_loss in (True, False):
        c0, r0, jp0, jl0, v0 = oracle.evaluate(w, o, apply_loss)
        c1, r1, jp1, jl1, v1 = ctx.evaluate(w, o, apply_loss)
        assert np.array_equal(v0, v1)
        assert abs(c0 - c1) <= 1e-12 * max(1.0, abs(c0))
        assert np.abs(r0 - r1).max() <= 1e-9 * max(1.0, np.abs(r0).max())
        assert np.abs(jp0 - jp1).max() <= 1e-9 * max(1.0, np.abs(jp0).max())
        assert np.abs(jl0 - jl1).max() <= 1e-9 * max(1.0, np.abs(jl0).max())
        # a row that does not exist is zero: the depth row of an observation without a depth measurement
        no_d = w.obs_d <= 0
        assert not r1.reshape(-1, 3)[no_d, 2].any() and not jp1.reshape(-1, 18)[no_d, 12:].any() and not jl1.reshape(-1, 9)[no_d, 6:].any()


def test_ground_and_regulariser_rows_match_oracle(ctx, oracle):
    """SURVEY §8 B3 / B4 per entry on the GPU: ground-plane height rows and every regulariser row (scale, normal / distance
    smoothness, plane motion, global normal, speed prior) with their tangent-space Jacobians, <= 1e-9 against the oracle's
    dual numbers, on windows that exercise each wiring rule (tests/rows_common.py)."""
    import rows_common

    rows_common.run(lambda w, o, pose_only, prior: ctx.evaluate_rows(w, o, pose_only, prior), oracle)


def test_solve_c1_reprojection_only(ctx, oracle):
    check_solve_parity(ctx, oracle, synth.config_c1(), default_options())


def test_solve_small_depth_groundplane(ctx, oracle):
    check_solve_parity(ctx, oracle, synth.make_window(11, n_kf=4, n_lm=300), default_options())


def test_solve_c2_headline(ctx, oracle):
    rg, ro, wg, wo = check_solve_parity(ctx, oracle, synth.config_c2(), default_options())
    gt = wg.meta["gt_pose"]
    assert np.abs(wg.kf_pose[:, 4:] - gt[:, 4:]).max() < 0.05  # converges to the synthetic ground truth


def test_batch_ragged_matches_single(ctx, oracle):
    ws = [synth.make_window(100 + i, n_kf=3 + (i % 4), n_lm=150 + 90 * i) for i in range(6)]
    o = default_options()
    b = ba.Batch(ctx, [w.copy() for w in ws])
    b.solve(o)
    reps = b.download()
    for w, wb, rb in zip(ws, b.windows, reps):
        wo = w.copy()
        ro, _ = oracle.solve(wo, o)
        assert abs(rb["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
        assert rel_pose_err(wb.kf_pose, wo.kf_pose) <= TOL
    # reset + re-solve reproduces the same result bit for bit (deterministic reductions)
    first = [w.kf_pose.copy() for w in b.windows]
    b.reset()
    b.solve(o)
    b.download()
    for a, w in zip(first, b.windows):
        assert np.array_equal(a, w.kf_pose)
    b.close()


def test_streaming_and_lock_step_schedules_give_the_same_bits(ctx, monkeypatch):
    """The same batch through the device-side scheduler (windows stream through slots, KBA_STREAM_MIN=1) and through the
    host-driven lock-step schedule (KBA_STREAM_MIN above the batch size): identical parameters and reports.  The batch mixes
    window shapes, trimming and non-trimming windows, and one window WITHOUT landmarks (regularisers only), whose keyframe
    bookkeeping runs through the per-window kernels alone."""
    from limo_amd.window import Window

    ws = [synth.make_window(300 + i, n_kf=3 + (i % 5), n_lm=(60, 150, 400, 900)[i % 4], ground_frac=(0.0, 0.2)[i % 2]) for i in range(23)]
    d = {name: getattr(ws[0], name).copy() for name, _ in Window.FIELDS}
    for k in ("lm_pos", "lm_weight", "lm_is_ground", "obs_kf", "obs_lm", "obs_cam", "obs_u", "obs_v", "obs_d"):
        d[k] = d[k][:0]
    ws.insert(7, Window(**d))
    o = default_options()
    results = []
    for k in ("KBA_NO_COOP_SOLVE", "KBA_NO_WG_SOLVE"):  # (the one-launch paths have their own tests below)
        monkeypatch.setenv(k, "1")
    for stream_min in ("1", "1000"):
        monkeypatch.setenv("KBA_STREAM_MIN", stream_min)
        b = ba.Batch(ctx, [w.copy() for w in ws])
        b.solve(o)
        reps = b.download()
        results.append((reps, [(w.kf_pose.copy(), w.kf_plane_dir.copy(), w.kf_plane_dist.copy(), w.lm_pos.copy()) for w in b.windows], [b.trimmed(i) for i in range(len(ws))]))
        b.close()
    (ra, pa, ta), (rb, pb, tb) = results
    for i in range(len(ws)):
        for key in ("termination", "iterations_total", "successful_steps", "n_trimmed_landmarks", "final_cost", "initial_cost"):
            assert ra[i][key] == rb[i][key], (i, key)
        assert all(np.array_equal(x, y) for x, y in zip(pa[i], pb[i])), i
        assert np.array_equal(ta[i], tb[i])
    assert ra[7]["iterations_total"] == 0 and ra[7]["termination"] == 0


def test_not_enough_keyframes(ctx):
    """Only a window without active keyframes is refused; two (or one) active keyframes are solved like the reference
    does (its NotEnoughKeyframesException counts PUSHED keyframes: the shim's check) - cases kf2 / kf1 below."""
    from limo_amd.window import Window

    w2 = synth.make_window(5, n_kf=3, n_lm=50)
    w = Window(kf_pose=np.zeros((0, 7)), kf_plane_dir=np.zeros((0, 3)), kf_plane_dist=np.zeros(0), kf_fixation=np.zeros(0, np.int32),
               cam=w2.cam, lm_pos=w2.lm_pos, lm_weight=w2.lm_weight, lm_is_ground=w2.lm_is_ground, obs_kf=np.zeros(0, np.int32),
               obs_lm=np.zeros(0, np.int32), obs_cam=np.zeros(0, np.int32), obs_u=np.zeros(0, np.float32),
               obs_v=np.zeros(0, np.float32), obs_d=np.zeros(0, np.float32))
    with pytest.raises(ba.NotEnoughKeyframes):
        ctx.solve(w, default_options())


def test_solver_time_cap(ctx, oracle):
    from test_emu_vs_oracle import check_time_cap

    last = {}

    def solve(w, o):
        b = ba.Batch(ctx, [w])
        b.solve(o)
        rep = b.download()[0]
        last["t"] = b.trimmed(0)
        b.close()
        return rep

    check_time_cap(solve, lambda w, o: oracle.solve(w, o)[0], lambda: last["t"], oracle.last_trimmed)


# ------------------------------------------------------------------------------------------------------------------
from test_emu_vs_oracle import CASES, make_pose_only_case  # noqa: E402  (same cases as the CPU tier)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "kf%d_lm%d_s%d" % (c["n_kf"], c["n_lm"], c["seed"]))
def test_solve_cases_match_oracle(ctx, oracle, case):
    kw = dict(case)
    w = synth.make_window(kw.pop("seed"), **kw)
    rg, ro, wg, wo = check_solve_parity(ctx, oracle, w, default_options())
    assert np.abs(wg.kf_plane_dist - wo.kf_plane_dist).max() <= 1e-4
    assert np.allclose(np.linalg.norm(wg.kf_pose[:, :4], axis=1), 1.0, atol=1e-12)


@pytest.mark.parametrize("with_prior", [False, True])
def test_pose_only_matches_oracle(ctx, oracle, with_prior):
    pw, prior, gt = make_pose_only_case(71)
    o = default_options(min_landmarks_for_trimming=30)
    pg, po = pw.copy(), pw.copy()
    rg = ctx.adjust_pose_only(pg, prior if with_prior else None, o)
    ro = oracle.adjust_pose_only(po, prior if with_prior else None, o)
    assert rg["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
    assert np.abs(pg.kf_pose - po.kf_pose).max() <= 1e-6
    assert np.array_equal(pg.lm_pos, pw.lm_pos)


@pytest.mark.parametrize("seed", [71, 72, 73, 74])
@pytest.mark.parametrize("cap", [-1.0, 20.0])
def test_pose_only_one_launch_equals_lock_step(ctx, seed, cap, monkeypatch):
    """adjustPoseOnly runs its whole solveTrimmed schedule in ONE launch (kba_kernels.hip:k_solve_wg: the device functions
    of the lock-step kernels behind workgroup barriers).  Same arithmetic, same summation orders: the result must be
    bit-identical to the launch-per-phase path (KBA_NO_WG_SOLVE=1), with and without a wall-clock cap, with and without
    the speed prior, trimming included."""
    pw, prior, _ = make_pose_only_case(seed)
    o = default_options(min_landmarks_for_trimming=30, max_solver_time_sec=cap)
    for pr in (None, prior):
        # three paths: one workgroup (k_solve_wg), the cooperative kernel with G = 1 (KBA_NO_WG_SOLVE=1), the lock-step launches
        runs = []
        for env in ({}, {"KBA_NO_WG_SOLVE": "1"}, {"KBA_NO_WG_SOLVE": "1", "KBA_NO_COOP_SOLVE": "1"}):
            for k in ("KBA_NO_WG_SOLVE", "KBA_NO_COOP_SOLVE"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            x = pw.copy()
            runs.append((x, ctx.adjust_pose_only(x, pr, o)))
        for k in ("KBA_NO_WG_SOLVE", "KBA_NO_COOP_SOLVE"):
            monkeypatch.delenv(k, raising=False)
        (a, ra) = runs[0]
        for b, rb in runs[1:]:
            assert a.kf_pose.tobytes() == b.kf_pose.tobytes()
            for k in ("final_cost", "initial_cost", "iterations_total", "iterations_final", "num_solves", "n_trimmed_landmarks", "termination",
                      "successful_steps", "num_linearizations"):
                assert ra[k] == rb[k], (k, ra[k], rb[k])
        assert ra["n_trimmed_landmarks"] > 0 and ra["num_solves"] >= 2  # the trimming branch ran


def _solve_both_paths(ctx, w, o, monkeypatch):
    a, b = w.copy(), w.copy()
    for k in ("KBA_NO_COOP_SOLVE", "KBA_NO_WG_SOLVE"):
        monkeypatch.delenv(k, raising=False)
    ra = ctx.solve(a, o)
    for k in ("KBA_NO_COOP_SOLVE", "KBA_NO_WG_SOLVE"):
        monkeypatch.setenv(k, "1")
    rb = ctx.solve(b, o)
    for k in ("KBA_NO_COOP_SOLVE", "KBA_NO_WG_SOLVE"):
        monkeypatch.delenv(k, raising=False)
    return a, ra, b, rb


@pytest.mark.parametrize("cap", [-1.0, 30.0])
def test_single_window_one_launch_equals_lock_step(ctx, cap, monkeypatch):
    """limo_ba_solve of one window is ONE cooperative launch (kba_kernels.hip:k_solve_coop: G workgroups that meet at
    device-wide barriers where the lock-step solve has launch boundaries; camera system and Schur complement side by side).
    Same device functions, partitions and summation orders: bit-identical to the launch-per-phase path
    (KBA_NO_COOP_SOLVE=1) - C2 windows, the reference-test shapes, ground plane on / off, stereo, trimming on / off,
    1 .. 4 free keyframes, a window without any landmark left."""
    from fuzz_common import random_windows

    o = default_options(max_solver_time_sec=cap)
    ws = [synth.make_window(3000 + i) for i in range(3)]
    ws += [synth.make_window(c["seed"], **{k: v for k, v in c.items() if k != "seed"}) for c in CASES]
    ws += [w for kw, w in random_windows(40, 2024) if kw["n_kf"] <= 5][:12]
    n_coop = 0
    for w in ws:
        a, ra, b, rb = _solve_both_paths(ctx, w, o, monkeypatch)
        assert a.kf_pose.tobytes() == b.kf_pose.tobytes()
        assert a.lm_pos.tobytes() == b.lm_pos.tobytes()
        assert a.kf_plane_dir.tobytes() == b.kf_plane_dir.tobytes() and a.kf_plane_dist.tobytes() == b.kf_plane_dist.tobytes()
        for k in ("final_cost", "initial_cost", "iterations_total", "iterations_final", "num_solves", "n_trimmed_landmarks", "termination",
                  "successful_steps", "num_linearizations"):
            assert ra[k] == rb[k], (k, ra[k], rb[k])
        n_coop += 1
    assert n_coop >= 10


def test_barrier_timeout_of_the_one_launch_solve_is_recovered(ctx, monkeypatch):
    """A device-wide barrier of k_solve_coop that is not met in time aborts the launch with poses / landmarks / LM state
    half-updated (the constant clock it waits on keeps running while a wave is preempted: shared GPU, debugger, profiler).
    limo_ba_batch_solve then restores the batch's initial state and takes the launch sequence: no error reaches the caller
    and the result is the lock-step result bit for bit.  KBA_COOP_TIMEOUT_MS=0 makes every barrier give up at its first wait."""
    o = default_options()
    for w in (synth.config_c2(), synth.make_window(3001)):
        for k in ("KBA_NO_COOP_SOLVE", "KBA_NO_WG_SOLVE", "KBA_COOP_TIMEOUT_MS"):
            monkeypatch.delenv(k, raising=False)
        ref = w.copy()
        monkeypatch.setenv("KBA_NO_COOP_SOLVE", "1")
        r_ref = ctx.solve(ref, o)
        monkeypatch.delenv("KBA_NO_COOP_SOLVE")
        before = ctx.coop_fallbacks()
        monkeypatch.setenv("KBA_COOP_TIMEOUT_MS", "0")
        x = w.copy()
        r = ctx.solve(x, o)
        monkeypatch.delenv("KBA_COOP_TIMEOUT_MS")
        assert ctx.coop_fallbacks() == before + 1            # the launch did give up ...
        assert x.kf_pose.tobytes() == ref.kf_pose.tobytes() and x.lm_pos.tobytes() == ref.lm_pos.tobytes()  # ... and nothing of it is left
        for k in ("final_cost", "initial_cost", "iterations_total", "num_solves", "n_trimmed_landmarks", "termination", "num_linearizations"):
            assert r[k] == r_ref[k], (k, r[k], r_ref[k])
        y = w.copy()
        ctx.solve(y, o)                                        # the next call on the context takes the one-launch path again
        assert ctx.coop_fallbacks() == before + 1 and y.kf_pose.tobytes() == ref.kf_pose.tobytes()


@pytest.mark.parametrize("n", [24, 64, 200])
def test_small_batch_one_launch_equals_streaming(ctx, n, monkeypatch):
    """Batches of up to 64 fast-class windows run as one cooperative launch with fewer workgroups per window (64 windows:
    4 each; with KBA_COOP_MAX_WIN=256 also 200 windows, 1 each): same bits as the streaming solve of the same batch
    (KBA_STREAM_MIN=1 asks for it)."""
    monkeypatch.setenv("KBA_COOP_MAX_WIN", "256")
    ws = [synth.make_window(5200 + i, n_kf=3 + (i % 3), n_lm=(150, 400, 900, 2000)[i % 4], ground_frac=(0.0, 0.2)[i % 2]) for i in range(n)]
    o = default_options()
    results = []
    for streaming in (False, True):
        monkeypatch.delenv("KBA_STREAM_MIN", raising=False)
        if streaming:
            monkeypatch.setenv("KBA_STREAM_MIN", "1")
        b = ba.Batch(ctx, [w.copy() for w in ws])
        b.solve(o)
        reps = b.download()
        results.append((reps, [(w.kf_pose.copy(), w.kf_plane_dir.copy(), w.kf_plane_dist.copy(), w.lm_pos.copy()) for w in b.windows], [b.trimmed(i) for i in range(len(ws))]))
        b.close()
    monkeypatch.delenv("KBA_STREAM_MIN", raising=False)
    (ra, pa, ta), (rb, pb, tb) = results
    for i in range(len(ws)):
        for key in ("termination", "iterations_total", "successful_steps", "n_trimmed_landmarks", "final_cost", "initial_cost", "num_solves"):
            assert ra[i][key] == rb[i][key], (i, key)
        assert all(np.array_equal(x, y) for x, y in zip(pa[i], pb[i])), i
        assert np.array_equal(ta[i], tb[i])


def test_committed_golden_fixtures(ctx):
    """GPU results against tests/golden/oracle_windows.json (oracle outputs committed with their generator), so the
    GPU tier has fixed targets that do not depend on the oracle being rebuilt on the GPU box."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_windows.json")) as f:
        gold = json.load(f)
    for g in gold["cases"]:
        w = synth.make_window(g["seed"], n_kf=g["n_kf"], n_lm=g["n_lm"], **g.get("kw", {}))
        rep = ctx.solve(w, default_options())
        assert rep["n_trimmed_landmarks"] == g["n_trimmed"]
        assert abs(rep["final_cost"] - g["final_cost"]) <= TOL * abs(g["final_cost"])
        assert rel_pose_err(w.kf_pose, np.array(g["kf_pose"])) <= TOL


def test_full_size_batch_properties(ctx):
    """BASELINE configs at full size through size-independent properties: a batch of C2 windows and one C4-sized
    window (10 keyframes x 8000 landmarks): every window terminates, the robust cost never increases, fixed
    keyframes stay bit-identical, quaternions / plane normals stay unit, the estimate moves towards ground truth,
    and solving a window alone or inside a batch gives the same bits."""
    ws = [synth.make_window(900 + i) for i in range(12)]
    o = default_options()
    b = ba.Batch(ctx, [w.copy() for w in ws])
    b.solve(o)
    reps = b.download()
    # the C4-sized window goes through the generic Schur kernel (more than four free keyframes), on its own
    c4 = synth.config_c4()
    c4_out = c4.copy()
    c4_rep = ctx.solve(c4_out, o)
    for w0, w1, r in zip(ws + [c4], b.windows + [c4_out], reps + [c4_rep]):
        assert r["termination"] in (0, 1)
        # initial_cost == -1: the first solve failed at x0 (a reprojection functor with |z| < 0.01); trimming then
        # removes that landmark and the final solve runs - same as the reference / oracle.
        assert r["initial_cost"] == -1.0 or r["final_cost"] < r["initial_cost"]
        assert np.array_equal(w0.kf_pose[0], w1.kf_pose[0])
        assert np.allclose(np.linalg.norm(w1.kf_pose[:, :4], axis=1), 1.0, atol=1e-12)
        assert np.allclose(np.linalg.norm(w1.kf_plane_dir, axis=1), 1.0, atol=1e-12)
        gt = w0.meta["gt_pose"]
        assert np.abs(w1.kf_pose[:, 4:] - gt[:, 4:]).max() < np.abs(w0.kf_pose[:, 4:] - gt[:, 4:]).max()
        assert 0 < r["n_trimmed_landmarks"] <= int(0.11 * w0.n_lm) + 1  # two 5 % lists, union
    alone = ws[3].copy()
    ctx.solve(alone, o)
    assert np.array_equal(alone.kf_pose, b.windows[3].kf_pose)
    b.close()


def test_invalid_input_is_rejected(ctx):
    w = synth.make_window(5, n_kf=3, n_lm=50)
    w.obs_kf[0] = 99
    with pytest.raises(ba.LimoError):
        ctx.solve(w, default_options())
    big = synth.make_window(6, n_kf=33, n_lm=40)  # more than kMaxKf keyframes
    with pytest.raises(ba.LimoError):
        ctx.solve(big, default_options())


def test_window_of_the_drive_with_a_failed_landmark_cholesky(ctx, oracle):
    """tests/golden/window_drive_frame1674.npz: the window `solve()` received at frame 1674 of the 4541-frame drive (dumped
    with LIMO_KBA_DUMP, tests/window_io.py).  Its final solve runs ~60 iterations with 13 rejected steps; at a large trust
    radius the damped 3x3 block of a weakly observed landmark loses positive definiteness - an INVALID STEP (shrink the
    radius, go on), which the landmark pass once reported as a failed Jacobian evaluation (a logical OR read bitwise): the
    solve ended with FAILURE where the oracle converges."""
    import os

    import window_io

    w = window_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_drive_frame1674.npz"))
    rg, ro, _, _ = check_solve_parity(ctx, oracle, w, default_options())
    assert ro["termination"] == 0 and rg["termination"] == 0
    assert rg["iterations_total"] > 40  # the long plateau is what makes the radius grow


def test_large_batches_and_the_contexts_pack_arena(ctx):
    """Batches of 128 windows or more are packed into a pinned host arena the context keeps (limo_ctx.hpp:pack_arena): made behind the
    first such batch, lent to ONE live batch at a time - a second large batch created while the first is alive packs into the heap -,
    recycled when its holder is destroyed, even with that batch's upload still in flight.  Whatever memory a batch was packed into,
    its results are the same bits."""
    o = default_options()
    ws = [synth.make_window(8100 + i, n_kf=3 + i % 3, n_lm=120 + 7 * (i % 40)) for i in range(160)]

    def solved(batch):
        batch.solve(o)
        batch.download()
        return [(w.kf_pose.tobytes(), w.lm_pos.tobytes()) for w in batch.windows]

    a = ba.Batch(ctx, [w.copy() for w in ws])       # (first large batch of this test: measures or holds the arena)
    b = ba.Batch(ctx, [w.copy() for w in ws])       # created while a is alive: heap
    ra, rb = solved(a), solved(b)
    assert ra == rb
    a.close()
    c = ba.Batch(ctx, [w.copy() for w in ws])       # the arena is free again
    c2 = ba.Batch(ctx, [w.copy() for w in ws[:130]])  # ... and taken: heap
    c.close()                                        # destroyed right behind its creation: its upload may still be in flight
    d = ba.Batch(ctx, [w.copy() for w in ws])       # packed over c's arrays
    rd, rc2 = solved(d), solved(c2)
    assert rd == ra and rc2 == ra[:130]
    for x in (b, c2, d):
        x.close()
