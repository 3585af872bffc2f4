"""The reference's own gtest scenarios (restated in tests/cpp/test_kba_shim.cpp) against the keyframe_bundle_adjustment
C++ shim (limo_amd/kba): CPU tier links the emulated C-ABI, GPU tier links liblimo_hip.so."""
import os
import subprocess

import numpy as np
import pytest

import emu_ffi


def run(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, "C++ shim tests failed:\n" + r.stdout[-4000:]
    assert "0 failed tests" in r.stdout


def test_reference_scenarios_with_emulated_backend():
    run(emu_ffi.build_shim_tests(gpu=False))


def test_reference_scenarios_with_the_oracle_behind_the_shim():
    """The same gtest scenarios with oracle/kba_oracle.cpp behind the C-ABI (tests/cpp/oracle_abi.cpp): the convergence scenes
    of test/keyframe_bundle_adjustment.cpp:794-858,1090-1145 hold for the restated Ceres loop as they do for the kernels."""
    run(emu_ffi.build_shim_tests(oracle=True))


def test_reference_scenarios_under_address_sanitizer():
    """The same scenarios under -fsanitize=address - among them Keyframe.measurementTable and
    BundleAdjusterKeyframes.solveFollowsMeasurementEdits, which erase and insert measurements through the public member between
    two uses of a keyframe: a table that outlived such an edit would read freed map nodes."""
    exe = emu_ffi.build_shim_tests_asan()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    print(r.stdout[-3000:])
    print(r.stderr[-3000:])
    assert r.returncode == 0 and "0 failed tests" in r.stdout and "AddressSanitizer" not in r.stderr


@pytest.mark.gpu
def test_reference_scenarios_on_gpu():
    run(emu_ffi.build_shim_tests(gpu=True))


def run_stream(exe, n_frames, n_lm, env=None, extra=()):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, str(n_frames), str(n_lm)] + list(extra), capture_output=True, text=True, timeout=900, env=e)
    print(r.stdout[-2000:])
    print(r.stderr[-1000:])
    assert r.returncode == 0, "streaming scenario failed:\n" + r.stdout[-2000:]
    assert " 0 failed" in r.stdout
    return r.stdout


def test_streaming_sequence_with_emulated_backend():
    """BASELINE.json configs[4] in miniature: sliding 5-keyframe window over a synthetic drive, the per-frame call
    order of the reference's node (adjustPoseOnly -> push -> deactivateKeyframes -> solve), ATE against ground truth."""
    exe = emu_ffi.build_stream_test(gpu=False)
    run_stream(exe, 24, 800)
    # the same drive heading backwards in the origin frame (quaternions of the poses in the trace <= 0 branch of the
    # matrix -> quaternion conversion): the accuracy thresholds hold for any heading
    run_stream(exe, 24, 800, env={"STREAM_YAW0": "2.6"})


@pytest.mark.gpu
def test_streaming_sequence_on_gpu():
    run_stream(emu_ffi.build_stream_test(gpu=True), 80, 2500)


@pytest.mark.gpu
def test_long_streaming_sequence_on_gpu_is_reproducible():
    """600 frames (the heading passes 180 degrees): every call must succeed, the drift stays small, and two runs of the
    same drive print the same per-frame trace - the device entry points are deterministic and leave nothing behind."""
    exe = emu_ffi.build_stream_test(gpu=True)
    outs = [run_stream(exe, 600, 18600, env={"STREAM_TRACE": "1"}, extra=["long"]) for _ in range(2)]
    frames = [[l for l in o.splitlines() if l.startswith("frame ")] for o in outs]
    assert len(frames[0]) == 600 and frames[0] == frames[1]
    summary = [l for l in outs[0].splitlines() if l.startswith("stream: 600 frames")][0]
    ate = float(summary.split("ATE rmse ")[1].split(" m")[0])
    assert ate < 0.25, summary
    assert "non-finite" not in outs[0]  # no landmark position is ever NaN / inf


# ---- apps/limo_stream: the product's streaming driver (limo_amd/kba/stream_driver.hpp) on the synthetic drive, LiDAR depth
#      assignment in the loop (BASELINE.json configs[4] end to end)
def run_limo_stream(exe, frames, az, poses_path=None, extra=()):
    cmd = [exe, "--frames", str(frames), "--az", str(az), "--quiet"] + list(extra)
    if poses_path:
        cmd += ["--poses", poses_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    print(r.stdout[-1500:])
    print(r.stderr[-800:])
    assert r.returncode == 0
    return {l.split()[0]: float(l.split()[1]) for l in r.stdout.splitlines() if len(l.split()) == 2 and l.split()[0] in ("frames", "fps", "ate_rmse", "ate_max", "depth_fraction", "keyframes", "solves", "solves_on_non_keyframes", "depth_prefetched")}


def test_limo_stream_with_emulated_backend(tmp_path):
    """Sweep -> depth assignment -> FeaturePoint::d -> adjustPoseOnly -> keyframe selection -> push -> deactivateKeyframes
    -> solve -> KITTI pose rows, on the emulated C-ABI (BA) + the oracle's depth assignment."""
    exe = emu_ffi.build_stream_app(gpu=False)
    poses = str(tmp_path / "poses.txt")
    out = run_limo_stream(exe, 40, 2000, poses)
    assert out["frames"] == 40 and out["keyframes"] >= 15 and out["solves"] >= 12
    # the node solves every time_between_keyframes on whatever frame arrives, keyframe or not (mono_lidar.cpp:245-246): with
    # a flow threshold that rejects most frames as keyframes the solves go on between them
    sparse = run_limo_stream(exe, 40, 2000, extra=["--min-flow", "30"])
    assert sparse["keyframes"] < out["keyframes"] and sparse["solves_on_non_keyframes"] >= 5 and sparse["solves"] > sparse["keyframes"]
    assert sparse["ate_rmse"] < 0.1
    assert out["depth_fraction"] > 0.3          # the features' depths come from the sweep, nowhere else
    assert out["ate_rmse"] < 0.05 and out["ate_max"] < 0.12
    rows = [l.split() for l in open(poses).read().splitlines() if l.strip()]
    assert len(rows) == 40 and all(len(r) == 12 for r in rows)  # KITTI odometry format, mono_lidar.cpp:281-294
    # the pose rows of this drive are committed (tests/golden/, written by this very command): every decision of the host
    # side - keyframe and landmark selection, window cut, flattening - and the emulated arithmetic behind the C-ABI are pinned
    # against drift; the rewrites of the selector's inner loops had to leave these rows byte-identical (LIMO_WRITE_GOLDEN=1
    # rewrites the file after an INTENDED change of behaviour)
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "limo_stream_emulated_40_frames_poses.txt")
    if os.environ.get("LIMO_WRITE_GOLDEN"):
        import shutil

        shutil.copy(poses, golden)
    want = np.loadtxt(golden)
    got = np.array(rows, float)
    assert want.shape == got.shape and np.abs(want - got).max() <= 1e-9, np.abs(want - got).max()
    first = np.array(rows[0], float).reshape(3, 4)
    assert np.allclose(first, np.eye(4)[:3], atol=1e-12)        # the first frame is the origin
    # the driver assigned the depths of every frame but the first one frame ahead (StreamDriver::announceNextFrame: on its depth
    # thread; --depth-ahead stream: limo_depth_estimate_begin / _end on the calling thread); assigned inside each frame's own
    # process() call instead (none), the rows are the same bytes
    assert out["depth_prefetched"] == 39  # (default: the driver's depth thread)
    for mode, n_ahead in (("stream", 39), ("none", 0)):
        other = str(tmp_path / ("poses_%s.txt" % mode))
        assert run_limo_stream(exe, 40, 2000, other, extra=["--depth-ahead", mode])["depth_prefetched"] == n_ahead
        assert open(other).read() == open(poses).read()
    # a frame announced that is NOT the one process() gets next (every third call announces the current frame again): what was
    # prepared is recognised as foreign (stamp / counts), dropped, the frame's depths are assigned in its own call - same bytes
    for mode in ("thread", "stream"):
        other = str(tmp_path / ("poses_mis_%s.txt" % mode))
        mis = run_limo_stream(exe, 40, 2000, other, extra=["--depth-ahead", mode, "--misannounce-every", "3"])
        assert 0 < mis["depth_prefetched"] < 39 and open(other).read() == open(poses).read()
    no_depth = run_limo_stream(exe, 40, 2000, extra=["--no-depth"])
    assert no_depth["depth_fraction"] == 0.0 and no_depth["ate_rmse"] > out["ate_rmse"]  # monocular: scale drifts without LiDAR
    # the node's own prior when it has no tf: five-point direction + the last keyframes' speed (mono_lidar.cpp:157-186,
    # limo_amd/kba/five_point.hpp) instead of constant velocity - adjustPoseOnly refines either, the drive stays on track
    fp = run_limo_stream(exe, 40, 2000, extra=["--five-point-prior"])
    assert fp["frames"] == 40 and fp["solves"] >= 12 and fp["ate_rmse"] < 0.1 and fp["ate_max"] < 0.6  # (worst: frames before the first solve, prior unrefined)


def test_limo_stream_replays_velodyne_scans(tmp_path):
    """KITTI velodyne scans written by one run (limo_amd/kba/kitti_io.hpp) and replayed by the next give the same pose
    rows; the ground-truth file has the same 12-column format."""
    exe = emu_ffi.build_stream_app(gpu=False)
    scans = tmp_path / "velodyne"
    scans.mkdir()
    a, b, gt = str(tmp_path / "a.txt"), str(tmp_path / "b.txt"), str(tmp_path / "gt.txt")
    run_limo_stream(exe, 10, 2000, a, extra=["--dump-velodyne", str(scans), "--gt-poses", gt])
    files = sorted(os.listdir(scans))
    assert files[0] == "000000.bin" and len(files) == 10
    cloud = np.fromfile(scans / "000003.bin", np.float32).reshape(-1, 4)
    assert 50000 < cloud.shape[0] < 200000 and np.isfinite(cloud).all()
    run_limo_stream(exe, 10, 2000, b, extra=["--velodyne", str(scans)])
    assert open(a).read() == open(b).read()
    g = np.loadtxt(gt)
    assert g.shape == (10, 12) and np.allclose(g[0], np.eye(4)[:3].ravel())
    assert 4.5 < g[9, 11] < 5.5  # 9 frames x 0.55 m along the camera's z axis


@pytest.mark.gpu
def test_limo_stream_on_gpu_matches_the_emulated_drive(tmp_path):
    """The same drive through liblimo_hip.so (depth.hip + the BA kernels) and through the emulated / oracle backend:
    trajectory parity frame by frame, plus accuracy against the ground truth."""
    gpu, emu = emu_ffi.build_stream_app(gpu=True), emu_ffi.build_stream_app(gpu=False)
    pg, pe = str(tmp_path / "gpu.txt"), str(tmp_path / "emu.txt")
    og = run_limo_stream(gpu, 80, 2000, pg)
    oe = run_limo_stream(emu, 80, 2000, pe)
    assert og["ate_rmse"] < 0.05 and og["depth_fraction"] > 0.35 and og["depth_prefetched"] == 79
    for mode, n_ahead in (("stream", 79), ("none", 0)):  # the depth thread (default), begin / _end on the calling thread, no look-ahead: the same bytes
        ps = str(tmp_path / ("gpu_%s.txt" % mode))
        assert run_limo_stream(gpu, 80, 2000, ps, extra=["--depth-ahead", mode])["depth_prefetched"] == n_ahead and open(ps).read() == open(pg).read()
    assert og["depth_fraction"] == oe["depth_fraction"]  # the depth assignment is bit-exact against its oracle (test_depth.py)
    a = np.array([l.split() for l in open(pg).read().splitlines() if l.strip()], float)
    b = np.array([l.split() for l in open(pe).read().splitlines() if l.strip()], float)
    assert a.shape == b.shape == (80, 12)
    # identical depths go into both drives; what is left is the arithmetic of the two BA implementations (block summation
    # orders), fed back through selection and trimming over 80 frames: 1 mm on a 44 m path
    assert np.abs(a[:, [3, 7, 11]] - b[:, [3, 7, 11]]).max() < 1e-3 and abs(og["ate_rmse"] - oe["ate_rmse"]) < 1e-3


def run_traced(exe, frames, poses_path):
    e = dict(os.environ, LIMO_STREAM_TRACE="1")
    r = subprocess.run([exe, "--frames", str(frames), "--az", "2000", "--quiet", "--poses", poses_path], capture_output=True, text=True, timeout=3000, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def check_against_oracle_drive(exe, frames, tmp_path, tag):
    """Config 5 against the ORACLE: the drive through `exe` and the same drive with oracle/kba_oracle.cpp, exact_oracle.cpp and
    depth_oracle.cpp behind every C-ABI call (tests/cpp/oracle_abi.cpp; call order mono_lidar.cpp:186-260).  Call by call
    (adjustPoseOnly and solve) the vehicle position agrees within 1e-4 x path travelled (north_star's pose bar) for as long
    as both drives select the same landmark sets; the first call whose selection differs is reported."""
    import stream_compare

    orc = emu_ffi.build_stream_app(oracle=True)
    pa, po = str(tmp_path / (tag + ".txt")), str(tmp_path / "oracle.txt")
    res = stream_compare.compare(run_traced(exe, frames, pa), run_traced(orc, frames, po))
    print("%s vs oracle drive, %d frames: %s" % (tag, frames, res))
    n_before = res["n_calls"] if res["first_divergence"] is None else res["first_divergence"]
    assert n_before >= min(res["n_calls"], 100), res    # the comparison must cover a real stretch of the drive
    assert res["max_rel_before"] <= 1e-4 and res["max_rel_cost_before"] <= 1e-4, res
    a, b = np.loadtxt(pa), np.loadtxt(po)
    assert a.shape == b.shape == (frames, 12)
    if res["first_divergence"] is None:  # same windows all the way: the trajectories ARE the same to rounding
        assert np.abs(a[:, [3, 7, 11]] - b[:, [3, 7, 11]]).max() <= 1e-4 * max(1.0, res["path_before"]), res
    return res


def test_limo_stream_emulated_drive_matches_the_oracle_drive(tmp_path):
    """CPU tier: the emulated kernels (same lane functions as the HIP binary) against the oracle drive, 80 frames."""
    res = check_against_oracle_drive(emu_ffi.build_stream_app(gpu=False), 80, tmp_path, "emulated")
    assert res["first_divergence"] is None and res["max_abs_before"] < 1e-6, res


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [80, 600])
def test_limo_stream_on_gpu_matches_the_oracle_drive(tmp_path, frames):
    """GPU tier: liblimo_hip.so (depth.hip, landmark_init.hip, k_solve_coop / k_solve_wg) against the oracle drive - NOT against
    the emulator that shares its source: 80 frames and 600 frames (330 m, ~900 C-ABI calls)."""
    res = check_against_oracle_drive(emu_ffi.build_stream_app(gpu=True), frames, tmp_path, "gpu")
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")  # record for profiles/
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "oracle_drive_%d.txt" % frames), "w") as f:
        f.write(repr(res) + "\n")


def test_stream_replicas_launcher_with_emulated_backend():
    """scripts/stream_replicas.py (config 5 on N GPUs = N independent sequences, one process each): two sequences on the
    emulated backend; each reports its own fps / ATE, different seeds give different drives."""
    import json
    import sys

    exe = emu_ffi.build_stream_app(gpu=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stream_replicas.py"), "--gpus", "2", "--frames", "12", "--exe", exe],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and len(out["sequences"]) == 2
    assert all(s["frames"] == 12 and s["ate_rmse"] < 0.05 for s in out["sequences"])
    assert out["sequences"][0]["ate_rmse"] != out["sequences"][1]["ate_rmse"]
