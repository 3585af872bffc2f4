"""KITTI file formats and trajectory errors."""
import numpy as np

from limo_amd import kitti_io, synth, synth_lidar


def test_velodyne_roundtrip(tmp_path):
    cloud = synth_lidar.make_sweep(3)[:5000]
    p = tmp_path / "000000.bin"
    kitti_io.write_velodyne_bin(p, cloud)
    assert p.stat().st_size == cloud.shape[0] * 16  # float32 x, y, z, intensity, no header
    back = kitti_io.read_velodyne_bin(p)
    assert back.dtype == np.float32 and np.array_equal(back, cloud.astype(np.float32))


def test_pose_file_roundtrip_and_conversion(tmp_path):
    w = synth.make_window(3, n_kf=4, n_lm=50)
    cam = synth.kitti_camera()[3:10]
    Ts = [kitti_io.keyframe_pose_to_camera_pose(p, cam) for p in w.meta["gt_pose"]]
    assert np.allclose(Ts[0], np.eye(4), atol=1e-12)  # keyframe 0 is the origin: the camera pose is the identity
    # the camera moves along its own +z (forward) when the vehicle drives along +x
    assert Ts[-1][2, 3] > 5.0 and abs(Ts[-1][0, 3]) < 1.0
    f = tmp_path / "poses.txt"
    kitti_io.write_kitti_poses(f, Ts)
    lines = f.read_text().strip().split("\n")
    assert len(lines) == 4 and all(len(l.split()) == 12 for l in lines)
    assert np.allclose(kitti_io.read_kitti_poses(f), np.array(Ts), atol=1e-10)


def test_trajectory_errors():
    rng = np.random.default_rng(0)
    ref = []
    T = np.eye(4)
    for _ in range(30):
        ref.append(T.copy())
        step = np.eye(4)
        step[:3, 3] = [0.0, 0.0, 1.0]
        c, s = np.cos(0.02), np.sin(0.02)
        step[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
        T = T @ step
    ref = np.array(ref)
    assert kitti_io.ate_rmse(ref, ref) == 0.0
    te, re_ = kitti_io.rpe(ref, ref)
    assert te < 1e-12 and re_ < 1e-7
    # a rigidly displaced copy: large ATE without alignment, zero with it; RPE is invariant
    G = np.eye(4)
    G[:3, 3] = [3.0, -1.0, 2.0]
    c, s = np.cos(0.3), np.sin(0.3)
    G[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    moved = np.array([G @ T for T in ref])
    assert kitti_io.ate_rmse(moved, ref) > 1.0
    assert kitti_io.ate_rmse(moved, ref, align=True) < 1e-9
    te, re_ = kitti_io.rpe(moved, ref)
    assert te < 1e-9
    noisy = ref.copy()
    noisy[:, :3, 3] += rng.normal(0, 0.05, (30, 3))
    assert 0.03 < kitti_io.ate_rmse(noisy, ref) < 0.2
