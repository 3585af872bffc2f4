"""CPU-tier tests of the product's HOST logic and KERNEL ARITHMETIC: the serial emulation (tests/cpp/emu_pipeline.cpp)
executes the same packing, LM state machine and per-lane statements as the gfx950 kernels and must agree with the
oracle (an independent dual-number / block-sparse implementation) on the same bytes."""
import numpy as np
import pytest

from limo_amd import _ffi, default_options, synth

TOL = 1e-4


def rel_pose_err(a, b):
    return np.abs(a[:, 4:] - b[:, 4:]).max() / max(1e-12, np.abs(b[:, 4:]).max())


@pytest.mark.parametrize("apply_loss", [False, True])
def test_analytic_jacobians_match_dual_numbers(oracle, emu, apply_loss):
    w = synth.make_window(31, n_kf=5, n_lm=400)
    o = default_options()
    c0, r0, jp0, jl0, v0 = oracle.evaluate(w, o, apply_loss)
    c1, r1, jp1, jl1, v1 = emu.evaluate(w, o, apply_loss)
    assert np.array_equal(v0, v1)
    assert abs(c0 - c1) <= 1e-12 * abs(c0)
    assert np.abs(r0 - r1).max() <= 1e-9 * max(1.0, np.abs(r0).max())
    assert np.abs(jp0 - jp1).max() <= 1e-10 * np.abs(jp0).max()
    assert np.abs(jl0 - jl1).max() <= 1e-10 * np.abs(jl0).max()


CASES = [
    dict(seed=1, n_kf=3, n_lm=200, depth_prob=0.0, ground_frac=0.0, with_ground_plane=False),  # C1
    dict(seed=41, n_kf=3, n_lm=80),  # below the trimming threshold (<= 100 landmarks)
    dict(seed=42, n_kf=4, n_lm=300),
    dict(seed=43, n_kf=6, n_lm=350, depth_prob=0.02),  # < 10 depth blocks per ... exercises plane-distance fixing
    dict(seed=44, n_kf=5, n_lm=250, ground_frac=0.0),  # depth but no ground landmarks
    dict(seed=45, n_kf=12, n_lm=200),  # the KITTI launch's window (max_size_optimization_window = 12)
    dict(seed=47, n_kf=2, n_lm=150),  # TWO active keyframes (deactivateKeyframes(min_conn, 3, max) can leave exactly these,
                                      # mono_lidar.cpp:249; the reference only refuses keyframes_.size() < 3 pushed ones)
    dict(seed=48, n_kf=1, n_lm=120),  # ONE active keyframe, Pose-fixed: landmarks only
    dict(seed=49, n_kf=20, n_lm=300),  # the reference's default max_size_optimization_window (bundle_adjuster_keyframes.hpp:129):
                                       # 190 free camera slots - k_schur_wide, camera system in global memory
    dict(seed=50, n_kf=14, n_lm=250, stereo_baseline=0.54),  # beyond the LDS-resident camera system, two cameras
    dict(seed=46, n_kf=4, n_lm=400, stereo_baseline=0.54),  # two cameras per keyframe (generic Schur path)
    dict(seed=909, n_kf=5, n_lm=2000),  # first solve FAILS at x0 (a reprojection block with |z| < 0.01), trimming removes it
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "kf%d_lm%d_s%d" % (c["n_kf"], c["n_lm"], c["seed"]))
def test_solve_matches_oracle(oracle, emu, case):
    kw = dict(case)
    w = synth.make_window(kw.pop("seed"), **kw)
    o = default_options()
    we, wo = w.copy(), w.copy()
    re_ = emu.solve_batch([we], o)[0]
    ro, _ = oracle.solve(wo, o)
    for k in ("n_depth_blocks", "n_repr_blocks", "n_gp_blocks", "n_trimmed_landmarks", "num_solves"):
        assert re_[k] == ro[k], k
    assert abs(re_["initial_cost"] - ro["initial_cost"]) <= 1e-10 * abs(ro["initial_cost"])
    assert abs(re_["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
    assert rel_pose_err(we.kf_pose, wo.kf_pose) <= TOL
    assert np.abs(we.kf_plane_dist - wo.kf_plane_dist).max() <= 1e-4
    assert np.array_equal(we.kf_pose[0], w.kf_pose[0])  # FixationStatus::Pose keyframe untouched
    assert np.allclose(np.linalg.norm(we.kf_pose[:, :4], axis=1), 1.0, atol=1e-12)
    assert np.allclose(np.linalg.norm(we.kf_plane_dir, axis=1), 1.0, atol=1e-12)


def test_trimming_removes_the_same_landmarks(oracle, emu):
    """Landmarks removed by trimming keep their last value (Ceres leaves removed blocks untouched), and both
    implementations remove the same set: compare which landmarks moved after the trimming round."""
    w = synth.make_window(51, n_kf=5, n_lm=400)
    o = default_options(max_num_iterations=0)  # final solve does no iteration: landmarks only move in the 2 trim iterations
    we, wo = w.copy(), w.copy()
    re_ = emu.solve_batch([we], o)[0]
    ro, _ = oracle.solve(wo, o)
    assert re_["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"] > 0
    assert np.allclose(we.lm_pos, wo.lm_pos, rtol=0, atol=1e-7)


def test_batch_is_independent_of_batch_composition(emu):
    ws = [synth.make_window(60 + i, n_kf=3 + i % 3, n_lm=120 + 60 * i) for i in range(4)]
    o = default_options()
    single = []
    for w in ws:
        c = w.copy()
        emu.solve_batch([c], o)
        single.append(c)
    batch = [w.copy() for w in ws]
    emu.solve_batch(batch, o)
    for a, b in zip(single, batch):
        assert np.array_equal(a.kf_pose, b.kf_pose) and np.array_equal(a.lm_pos, b.lm_pos)


def make_pose_only_case(seed):
    """New frame against fixed landmarks (adjustPoseOnly): limo_amd/synth.py:make_pose_only_case."""
    return synth.make_pose_only_case(seed)


@pytest.mark.parametrize("with_prior", [False, True])
def test_pose_only_matches_oracle(oracle, emu, with_prior):
    pw, prior, gt = make_pose_only_case(71)
    o = default_options(min_landmarks_for_trimming=30)  # adjustPoseOnly trims when selected > 30 (:865)
    pe, po = pw.copy(), pw.copy()
    re_ = emu.solve_batch([pe], o, pose_only=True, prior=prior if with_prior else None)[0]
    ro = oracle.adjust_pose_only(po, prior if with_prior else None, o)
    assert re_["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"]
    assert abs(re_["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
    assert np.abs(pe.kf_pose - po.kf_pose).max() <= 1e-6
    assert np.abs(pe.kf_pose[0, 4:] - gt[4:]).max() < 0.05  # pulled back to the true pose
    assert np.array_equal(pe.lm_pos, pw.lm_pos)  # landmarks are constant in motion-only adjustment


def test_not_enough_keyframes_and_bad_input(emu):
    import ctypes as C

    from limo_amd.window import struct_array

    lib = emu.load()
    from limo_amd.window import Window

    w2 = synth.make_window(5, n_kf=3, n_lm=50)
    z = np.zeros(0)
    w = Window(kf_pose=np.zeros((0, 7)), kf_plane_dir=np.zeros((0, 3)), kf_plane_dist=z, kf_fixation=np.zeros(0, np.int32), cam=w2.cam,
               lm_pos=w2.lm_pos, lm_weight=w2.lm_weight, lm_is_ground=w2.lm_is_ground, obs_kf=np.zeros(0, np.int32),
               obs_lm=np.zeros(0, np.int32), obs_cam=np.zeros(0, np.int32), obs_u=np.zeros(0, np.float32),
               obs_v=np.zeros(0, np.float32), obs_d=np.zeros(0, np.float32))
    arr = struct_array([w])
    o = default_options()
    assert lib.emu_ba_solve_batch(1, arr, C.byref(o), None, 0, None) == _ffi.LIMO_ERR_NOT_ENOUGH_KF  # no active keyframe at all
    w3 = synth.make_window(5, n_kf=3, n_lm=50)
    w3.obs_lm[0] = 10**6  # index out of range must be rejected, not read
    arr = struct_array([w3])
    assert lib.emu_ba_solve_batch(1, arr, C.byref(o), None, 0, None) == _ffi.LIMO_ERR_INVALID


def test_empty_and_ragged_windows(oracle, emu):
    """A window without observations / with unobserved landmarks must not crash and must leave parameters alone."""
    w = synth.make_window(81, n_kf=3, n_lm=30)
    from limo_amd.window import Window

    # no observations at all: only ground-plane rows of the ground landmarks + regularisers remain (the reference
    # builds those regardless of measurements, bundle_adjuster_keyframes.cpp:517-562) -> compare with the oracle
    empty = Window(**{n: getattr(w, n)[:0] if n.startswith("obs_") else getattr(w, n) for n, _ in Window.FIELDS})
    ee, eo = empty.copy(), empty.copy()
    rep = emu.solve_batch([ee], default_options())[0]
    ro, _ = oracle.solve(eo, default_options())
    assert rep["n_repr_blocks"] == 0 and rep["n_gp_blocks"] == ro["n_gp_blocks"]
    assert abs(rep["final_cost"] - ro["final_cost"]) <= 1e-6 * max(1e-12, abs(ro["final_cost"])) + 1e-12
    # no landmarks either: the only block is the scale regulariser with zero residual -> nothing moves
    bare = Window(**{n: (getattr(w, n)[:0] if (n.startswith("obs_") or n.startswith("lm_")) else getattr(w, n)) for n, _ in Window.FIELDS})
    before = bare.kf_pose.copy()
    rep = emu.solve_batch([bare], default_options())[0]
    assert np.array_equal(bare.kf_pose, before) and rep["termination"] == _ffi.LIMO_CONVERGENCE
    # drop all observations of one landmark: it is not part of the problem and keeps its value
    keep = w.obs_lm != 3
    rag = Window(**{n: getattr(w, n)[keep] if n.startswith("obs_") else getattr(w, n) for n, _ in Window.FIELDS})
    re_, ro = rag.copy(), rag.copy()
    emu.solve_batch([re_], default_options())
    oracle.solve(ro, default_options())
    assert np.array_equal(re_.lm_pos[3], rag.lm_pos[3])
    assert rel_pose_err(re_.kf_pose, ro.kf_pose) <= TOL


def test_random_window_shapes_match_oracle(oracle, emu):
    """A seeded sweep over window shapes the fixed cases above do not name (keyframe count, landmark count, share of
    depth / ground-plane / outlier measurements, mono / stereo, with / without plane parameters): the same trimming
    sets and terminations as the oracle, poses and cost inside the parity bar; solved one by one and as ONE ragged batch
    (the batch must give the bits of the single solves)."""
    rng = np.random.default_rng(2024)
    o = default_options()
    ws = []
    for i in range(10):
        kw = dict(n_kf=int(rng.integers(3, 9)), n_lm=int(rng.choice([40, 90, 150, 260, 420])),
                  depth_prob=float(rng.choice([0.0, 0.02, 0.3, 0.9])), ground_frac=float(rng.choice([0.0, 0.1, 0.4])),
                  outlier_frac=float(rng.choice([0.0, 0.05, 0.15])), stereo_baseline=float(rng.choice([0.0, 0.0, 0.54])),
                  with_ground_plane=bool(rng.integers(0, 2)))
        ws.append(synth.make_window(20000 + i, **kw))
    singles = []
    for w in ws:
        we, wo = w.copy(), w.copy()
        re_ = emu.solve_batch([we], o)[0]
        ro, _ = oracle.solve(wo, o)
        for k in ("n_depth_blocks", "n_repr_blocks", "n_gp_blocks", "n_trimmed_landmarks", "num_solves", "termination"):
            assert re_[k] == ro[k], k
        assert abs(re_["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
        assert rel_pose_err(we.kf_pose, wo.kf_pose) <= TOL
        singles.append(we)
    batch = [w.copy() for w in ws]
    emu.solve_batch(batch, o)
    for a, b in zip(batch, singles):
        assert np.array_equal(a.kf_pose, b.kf_pose) and np.array_equal(a.lm_pos, b.lm_pos)


def check_time_cap(solve, solve_oracle, last_trimmed, last_trimmed_oracle):
    """max_solver_time_sec (the reference runs with 0.15-0.2 s live and 20 s in its tests; robust_solving.hpp:104,
    Ceres checks the clock when an iteration is finalised).  A cap that has expired by the time iteration 0 is
    finalised stops every solve of the schedule before its first step: parameters untouched, NO_CONVERGENCE, zero LM
    iterations, trimming (evaluated at x0) identical to the oracle's; a generous cap changes nothing."""
    w = synth.make_window(52, n_kf=4, n_lm=300)
    tiny = default_options(max_solver_time_sec=1e-9)
    a, b = w.copy(), w.copy()
    ra = solve(a, tiny)
    ta = last_trimmed()
    rb = solve_oracle(b, tiny)
    tb = last_trimmed_oracle()
    for r in (ra, rb):
        assert r["termination"] == _ffi.LIMO_NO_CONVERGENCE and r["iterations_total"] == 0
    assert np.array_equal(a.kf_pose, w.kf_pose) and np.array_equal(a.lm_pos, w.lm_pos)
    assert np.array_equal(b.kf_pose, w.kf_pose) and np.array_equal(b.lm_pos, w.lm_pos)
    assert ra["n_trimmed_landmarks"] == rb["n_trimmed_landmarks"] > 0 and np.array_equal(ta, tb)
    assert abs(ra["final_cost"] - rb["final_cost"]) <= 1e-10 * abs(rb["final_cost"])
    free, capped = w.copy(), w.copy()
    rf = solve(free, default_options())
    rc = solve(capped, default_options(max_solver_time_sec=20.0))  # the value of the reference's own tests
    assert np.array_equal(free.kf_pose, capped.kf_pose) and np.array_equal(free.lm_pos, capped.lm_pos)
    assert rf["iterations_total"] == rc["iterations_total"] > 0 and rc["termination"] == _ffi.LIMO_CONVERGENCE


def test_solver_time_cap(oracle, emu):
    check_time_cap(lambda w, o: emu.solve_batch([w], o)[0], lambda w, o: oracle.solve(w, o)[0], lambda: emu.last_trimmed(0), oracle.last_trimmed)


def test_streaming_schedule_gives_the_lock_step_results(emu):
    """Streaming solve (windows move through slots, every window advances through solveTrimmed's phases on its own;
    kba_lm.hpp:sched_advance, kba_kernels.hip:k_sched) against the lock-step schedule (kba_pack.cpp:run_schedule): the
    same bits per window whatever the number of slots, and the same reports."""
    ws = [synth.make_window(300 + i, n_kf=3 + (i % 4), n_lm=[60, 150, 400, 900][i % 4], outlier_frac=0.05 * (i % 3)) for i in range(10)]
    o = default_options()
    ref = [w.copy() for w in ws]
    r_ref = emu.solve_batch(ref, o)
    for n_slots in (1, 3, 16):
        got = [w.copy() for w in ws]
        r_got = emu.solve_batch_streaming(got, o, n_slots)
        for a, b, ra, rb in zip(ref, got, r_ref, r_got):
            assert np.array_equal(a.kf_pose, b.kf_pose) and np.array_equal(a.lm_pos, b.lm_pos) and np.array_equal(a.kf_plane_dist, b.kf_plane_dist)
            for k in ("termination", "num_solves", "iterations_total", "iterations_final", "successful_steps", "n_trimmed_landmarks", "num_linearizations", "initial_cost", "final_cost"):
                assert ra[k] == rb[k], k
    two = default_options(num_trim_rounds=2)  # more than one trimming round
    a, b = [w.copy() for w in ws[:4]], [w.copy() for w in ws[:4]]
    ra, rb = emu.solve_batch(a, two), emu.solve_batch_streaming(b, two, 2)
    for x, y, p, q in zip(a, b, ra, rb):
        assert np.array_equal(x.kf_pose, y.kf_pose) and p["n_trimmed_landmarks"] == q["n_trimmed_landmarks"] and p["num_solves"] == q["num_solves"]


def test_window_of_the_drive_with_rejected_steps_at_a_large_radius(oracle, emu):
    """The fixture of tests/test_gpu_ba.py::test_window_of_the_drive_with_a_failed_landmark_cholesky through the emulated
    pipeline: ~60 iterations, 13 rejected steps, same minimum as the oracle."""
    import os

    import window_io

    w = window_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_drive_frame1674.npz"))
    we, wo = w.copy(), w.copy()
    re = emu.solve_batch([we], default_options())[0]
    ro, _ = oracle.solve(wo, default_options())
    assert re["termination"] == 0 and ro["termination"] == 0
    assert re["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"]
    assert abs(re["final_cost"] - ro["final_cost"]) <= TOL * abs(ro["final_cost"])
    assert rel_pose_err(we.kf_pose, wo.kf_pose) <= TOL
    assert re["iterations_total"] - re["successful_steps"] >= 5


def test_closed_form_rotation_jacobian_is_the_chain_rule_form(emu):
    """M(q, p) = d(R(q) p)/d(delta) as the kernels of the solve form it, -2 [Rh(q) p]_x from R(q) and |q|^2 - 1, against the chain-rule statement
    (d(R p)/dq times the plus-Jacobian of the quaternion update) that the Problem::Evaluate path keeps - a polynomial identity,
    also for quaternions that are not of unit length (the update keeps |q|)."""
    import ctypes as C

    import emu_ffi

    lib = emu_ffi.load()
    dp = C.POINTER(C.c_double)
    lib.emu_rot_tangent_forms.argtypes = [dp, dp, dp, dp]
    lib.emu_rot_tangent_forms.restype = None
    rng = np.random.default_rng(11)
    for k in range(200):
        q = rng.normal(size=4)
        q *= (1.0 if k % 2 == 0 else 1.0 + 0.2 * rng.normal()) / np.linalg.norm(q)
        p = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
        a, b = np.zeros(9), np.zeros(9)
        lib.emu_rot_tangent_forms(q.ctypes.data_as(dp), p.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp))
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max())


def test_packing_from_two_host_threads_at_once(emu):
    """The pack keeps its host threads between calls (one parallel region at a time); a second host thread that packs at the same
    moment finds the pool busy and takes the spawn path.  Both must produce what a lone call produces."""
    import threading

    o = default_options()
    batches = [[synth.make_window(300 + 10 * k + i, n_kf=3 + i % 3, n_lm=150 + 40 * i) for i in range(12)] for k in range(2)]
    ref = []
    for b in batches:
        ws = [w.copy() for w in b]
        emu.solve_batch(ws, o)
        ref.append(ws)
    for _ in range(3):
        got = [None, None]

        def work(k):
            ws = [w.copy() for w in batches[k]]
            emu.solve_batch(ws, o)
            got[k] = ws

        ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for k in range(2):
            for a, b in zip(got[k], ref[k]):
                assert np.array_equal(a.kf_pose, b.kf_pose) and np.array_equal(a.lm_pos, b.lm_pos)
