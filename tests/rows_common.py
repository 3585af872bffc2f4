"""Shared by the CPU tier (emulated pipeline) and the GPU tier (liblimo_hip.so): per-entry parity of the residual rows that
are NOT reprojection / depth blocks - ground-plane height rows (cost_functors_ceres.hpp:358-385, wiring
bundle_adjuster_keyframes.cpp:517-562) and the regularisers (B4: cost_functors_ceres.hpp:224-250,300-438,507-555, wiring
bundle_adjuster_keyframes.cpp:704-728,769-818,835-853,890-904) - against the oracle's dual-number rows."""
import numpy as np

from limo_amd import _ffi, default_options, synth

TOL = 1e-9


def perturbed(w, seed):
    """Planes away from their defaults, so that every regulariser has a non-trivial residual."""
    rng = np.random.default_rng(seed)
    w = w.copy()
    d = w.kf_plane_dir + rng.normal(0, 0.05, w.kf_plane_dir.shape)
    w.kf_plane_dir[:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    w.kf_plane_dist[:] = w.kf_plane_dist + rng.normal(0, 0.1, w.kf_plane_dist.shape)
    return w


def cases():
    """(name, window, pose_only, prior, what the problem must contain)."""
    out = []
    # headline window: ~400 ground rows (>= 30: NO scale block), plane regularisers, plane distances free
    out.append(("c2_ground", perturbed(synth.config_c2(), 1), False, None, {"gp_min": 30, "scale": False, "planes": True}))
    # few ground rows (1 <= #gp < 30): scale block with weight 1000 / (#depth + #gp)
    out.append(("few_ground", perturbed(synth.make_window(21, n_kf=4, n_lm=120, ground_frac=0.1), 2), False, None, {"gp_min": 1, "gp_max": 29, "scale": True, "planes": True}))
    # no depth at all (#depth < 10): plane distances are constant (bundle_adjuster_keyframes.cpp:722-728); ground rows remain
    out.append(("no_depth", perturbed(synth.make_window(22, n_kf=5, n_lm=400, depth_prob=0.0), 3), False, None, {"gp_min": 30, "scale": False, "planes": True, "dist_fixed": True}))
    # no ground plane: the scale block alone (weight 1000 / #depth)
    out.append(("no_ground", synth.make_window(23, n_kf=3, n_lm=150, ground_frac=0.0, with_ground_plane=False), False, None, {"gp_max": 0, "scale": True, "planes": False}))
    # C1: reprojection only: scale block with weight 1000
    out.append(("c1", synth.config_c1(), False, None, {"gp_max": 0, "scale": True, "planes": False}))
    # twelve keyframes (window-level kernels outside LDS), stereo rig
    out.append(("kf12_stereo", perturbed(synth.make_window(24, n_kf=12, n_lm=800, stereo_baseline=0.54), 4), False, None, {"gp_min": 30, "scale": False, "planes": True}))
    # adjustPoseOnly with the speed prior (SpeedRegularizationVector2)
    pw, prior, _ = synth.make_pose_only_case(71)
    out.append(("pose_only_speed", pw, True, prior, {"speed": True}))
    return out


def compare(name, got, want, expect):
    kg = {tuple(r["key"]): r for r in got}
    kw = {tuple(r["key"]): r for r in want}
    assert len(kg) == len(got) and len(kw) == len(want), name  # keys are unique
    assert set(kg) == set(kw), (name, sorted(set(kg) ^ set(kw))[:5])
    kinds = [k[0] for k in kw]
    n_gp = kinds.count(_ffi.ROW_GROUND_HEIGHT)
    assert n_gp >= expect.get("gp_min", 0) and n_gp <= expect.get("gp_max", 10 ** 9), (name, n_gp)
    if "scale" in expect:
        assert (kinds.count(_ffi.ROW_SCALE) == 1) == expect["scale"], name
    if "planes" in expect:
        assert (kinds.count(_ffi.ROW_PLANE_MOTION) > 0) == expect["planes"], name
        assert (kinds.count(_ffi.ROW_GLOBAL_NORMAL) > 0) == expect["planes"], name
    if expect.get("speed"):
        assert kinds.count(_ffi.ROW_SPEED) == 3, name
    worst = 0.0
    for key, w in kw.items():
        g = kg[key]
        assert g["fixed"] == w["fixed"], (name, key)
        jmax = max(1e-300, np.abs(w["jac_kf"]).max(), np.abs(w["jac_lm"]).max())
        e = max(abs(g["r"] - w["r"]) / max(1.0, abs(w["r"])), np.abs(g["jac_kf"] - w["jac_kf"]).max() / jmax, np.abs(g["jac_lm"] - w["jac_lm"]).max() / jmax)
        assert e <= TOL, (name, key, e, g, w)
        assert abs(g["cost"] - w["cost"]) <= 1e-12 * max(1e-300, abs(w["cost"])) + 1e-300 or abs(g["cost"] - w["cost"]) <= 1e-9 * abs(w["cost"]), (name, key, g["cost"], w["cost"])
        worst = max(worst, e)
    if expect.get("dist_fixed"):  # every plane-distance block constant: the distance-difference rows are fixed-cost rows
        assert all(kw[k]["fixed"] == 1 for k in kw if k[0] == _ffi.ROW_DIST_DIFF), name
    return len(kw), n_gp, worst


def run(evaluate_rows, oracle):
    o = default_options()
    for name, w, pose_only, prior, expect in cases():
        if pose_only:
            o2 = default_options()
            o2.min_landmarks_for_trimming = 30
        else:
            o2 = o
        got = evaluate_rows(w, o2, pose_only, prior)
        want = oracle.evaluate_rows(w, o2, pose_only, prior)
        n, n_gp, worst = compare(name, got, want, expect)
        print("rows %-16s %4d rows (%3d ground), worst relative entry error %.2e" % (name, n, n_gp, worst))
