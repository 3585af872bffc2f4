"""KITTI odometry file formats around the hot path: the velodyne scan the depth estimator reads, the
pose file the reference node dumps, and the trajectory errors used to compare runs.

  read_velodyne_bin / write_velodyne_bin   float32 x, y, z, intensity per return, no header - what
                                           demo_keyframe_bundle_adjustment_meta/apps/main_program/utility.h:11-40 reads;
                                           the array goes straight into limo_depth_estimate (cloud_xyzi).
  pose_to_kitti_row / write_kitti_poses    one line per frame, the first three rows of the 4x4 camera pose
                                           origin<-camera, row-major, 12 numbers
                                           (keyframe_bundle_adjustment_ros_tool/src/commons/general_helpers.hpp:24-29);
                                           the node converts a keyframe pose with
                                           T_cam_veh * T_kf_origin^-1 * T_cam_veh^-1 (mono_lidar.cpp:281-294).
  ate_rmse / rpe                           absolute trajectory error (optionally after a rigid alignment) and the
                                           relative pose error over a fixed frame offset (KITTI devkit convention:
                                           translation error per metre, rotation error per metre).
"""
import numpy as np

from .synth import Rt_to_pose, pose_to_Rt


def read_velodyne_bin(path):
    a = np.fromfile(path, dtype=np.float32)
    if a.size % 4:
        raise ValueError("%s: size is not a multiple of 4 floats" % path)
    return a.reshape(-1, 4)


def write_velodyne_bin(path, cloud_xyzi):
    np.ascontiguousarray(cloud_xyzi, np.float32).reshape(-1, 4).tofile(path)


def _T(pose7):
    R, t = pose_to_Rt(np.asarray(pose7, np.float64))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def keyframe_pose_to_camera_pose(pose_kf_origin, pose_cam_veh):
    """origin<-camera 4x4 of a keyframe pose (keyframe<-origin, 7) given the extrinsic camera<-vehicle (7)."""
    Tc = _T(pose_cam_veh)
    return Tc @ np.linalg.inv(_T(pose_kf_origin)) @ np.linalg.inv(Tc)


def pose_to_kitti_row(T44):
    return " ".join("%.12g" % x for x in np.asarray(T44)[:3, :].reshape(-1))


def write_kitti_poses(path, poses_44):
    with open(path, "w") as f:
        for T in poses_44:
            f.write(pose_to_kitti_row(T) + "\n")


def read_kitti_poses(path):
    out = []
    with open(path) as f:
        for line in f:
            v = np.array(line.split(), np.float64)
            if v.size == 0:
                continue
            if v.size != 12:
                raise ValueError("%s: expected 12 numbers per line" % path)
            T = np.eye(4)
            T[:3, :] = v.reshape(3, 4)
            out.append(T)
    return np.array(out)


def _align_rigid(est, ref):
    """Rotation + translation (no scale) that best maps est points onto ref points (Kabsch)."""
    ce, cr = est.mean(0), ref.mean(0)
    H = (est - ce).T @ (ref - cr)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, cr - R @ ce


def ate_rmse(est_44, ref_44, align=False):
    pe = np.asarray(est_44)[:, :3, 3]
    pr = np.asarray(ref_44)[:, :3, 3]
    if align:
        R, t = _align_rigid(pe, pr)
        pe = pe @ R.T + t
    return float(np.sqrt(np.mean(np.sum((pe - pr) ** 2, axis=1))))


def rpe(est_44, ref_44, delta=1):
    """Mean relative pose error over pairs (i, i+delta): (translation error / path length, rotation error [rad] / path length)."""
    est_44, ref_44 = np.asarray(est_44), np.asarray(ref_44)
    te, re_ = [], []
    for i in range(len(ref_44) - delta):
        d_ref = np.linalg.inv(ref_44[i]) @ ref_44[i + delta]
        d_est = np.linalg.inv(est_44[i]) @ est_44[i + delta]
        E = np.linalg.inv(d_ref) @ d_est
        length = max(1e-12, float(np.linalg.norm(d_ref[:3, 3])))
        te.append(np.linalg.norm(E[:3, 3]) / length)
        re_.append(np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)) / length)
    return float(np.mean(te)), float(np.mean(re_))
