"""Seeded synthetic KITTI-shaped optimisation windows (SURVEY.md §8d "Synthetic inputs").

There is no network and no KITTI data in this environment, so the benchmark and the parity tests run on
synthetic windows with the geometry of KITTI odometry sequence 00: gray-left camera intrinsics, the
camera<-vehicle extrinsic of the reference's launch file
(demo_keyframe_bundle_adjustment_meta/launch/tf2_static_aliases_kitti.launch:29), keyframes 0.4 s apart
(keyframe_ba_monolid.launch:40) at ~10 m/s, ~45 % of the observations carrying LiDAR depth, 20 % ground
landmarks at height_over_ground = 0.31 m (keyframe_ba_monolid.launch:56), 5 % gross outliers, 10 % of the
landmarks with the shrubbery weight 0.9 (keyframe_ba_monolid.launch:46).  Landmark start values are produced
the way BundleAdjusterKeyframes::push does (depth back-projection, else midpoint triangulation;
bundle_adjuster_keyframes.cpp:289-382) and the default cheirality rejection is applied
(landmark_selection_scheme_cheirality.cpp:22-40).

Pure numpy; deterministic for a given seed.
"""
import numpy as np

from . import _ffi
from .window import Window

KITTI_F = 718.856
KITTI_CX = 607.1928
KITTI_CY = 185.2157
KITTI_W = 1241
KITTI_H = 376
HEIGHT_OVER_GROUND = 0.31


# ----------------------------------------------------------------------------- pose algebra (w,x,y,z,tx,ty,tz)
def quat_to_R(q):
    """Eigen's un-normalised polynomial toRotationMatrix (reference: internal/definitions.hpp:75-83)."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def R_to_quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0 branch-stable."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.empty(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def pose_to_Rt(p):
    return quat_to_R(np.asarray(p[..., :4])), np.asarray(p[..., 4:7])


def Rt_to_pose(R, t):
    return np.concatenate([R_to_quat(R), t])


def rot_vec(v):
    """Rodrigues."""
    th = np.linalg.norm(v)
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rot_axis(axis, a):
    v = np.zeros(3)
    v[axis] = a
    return rot_vec(v)


def kitti_camera():
    """[f, cx, cy, q(4), t(3)] of the KITTI-00 gray-left camera; extrinsic camera<-vehicle.

    The launch file publishes vehicle->camera as `1.08 0 1.35 yaw=-pi/2 pitch=0 roll=-pi/2`
    (p_vehicle = T p_camera, T = Tr * Rz(yaw) Ry(pitch) Rx(roll)); we store its inverse.
    """
    R_vc = rot_axis(2, -np.pi / 2) @ rot_axis(0, -np.pi / 2)
    t_vc = np.array([1.08, 0.0, 1.35])
    R_cv = R_vc.T
    t_cv = -R_cv @ t_vc
    return np.concatenate([[KITTI_F, KITTI_CX, KITTI_CY], Rt_to_pose(R_cv, t_cv)])


# ----------------------------------------------------------------------------- generator
def make_window(
    seed,
    n_kf=5,
    n_lm=2000,
    depth_prob=0.45,
    ground_frac=0.2,
    outlier_frac=0.05,
    uv_sigma=0.5,
    depth_sigma=0.05,
    rot_noise_deg=1.0,
    trans_noise=0.15,
    shrub_frac=0.1,
    with_ground_plane=True,
    dt=0.4,
    stereo_baseline=0.0,
):
    """One synthetic window.  Returns a Window whose .meta holds the ground truth (gt_pose, gt_lm).
    stereo_baseline > 0 adds a second camera that far to the right of the first one (two views per keyframe;
    reprojection-only measurements), the multi-camera case of the reference (keyframe.cpp:5-17,43-59)."""
    rng = np.random.default_rng(int(seed))
    cam = kitti_camera()
    f, cx, cy = cam[0], cam[1], cam[2]
    R_cv, t_cv = pose_to_Rt(cam[3:10])

    # --- trajectory (vehicle frame: x forward, y left, z up); origin = vehicle frame of keyframe 0
    speed = rng.uniform(8.0, 12.0)
    yaw_rate = rng.uniform(-0.1, 0.1)
    R_ok = [np.eye(3)]
    p_ok = [np.zeros(3)]
    yaw = 0.0
    for k in range(1, n_kf):
        v = speed * (1.0 + rng.normal(0, 0.02))
        p_next = p_ok[-1] + R_ok[-1] @ np.array([v * dt, 0.0, 0.0])
        p_next[2] = rng.normal(0, 0.01)
        yaw += yaw_rate * dt
        Rk = rot_axis(2, yaw) @ rot_axis(1, rng.normal(0, np.deg2rad(0.3))) @ rot_axis(0, rng.normal(0, np.deg2rad(0.3)))
        R_ok.append(Rk)
        p_ok.append(p_next)
    R_ok = np.array(R_ok)
    p_ok = np.array(p_ok)
    # keyframe <- origin
    R_ko = np.transpose(R_ok, (0, 2, 1))
    t_ko = -np.einsum("kij,kj->ki", R_ko, p_ok)
    gt_pose = np.array([Rt_to_pose(R_ko[k], t_ko[k]) for k in range(n_kf)])

    # --- landmarks, sampled in the frustum of the newest keyframe
    n_ground = int(round(n_lm * ground_frac)) if with_ground_plane else 0
    n_free = n_lm - n_ground
    z = np.exp(rng.uniform(np.log(4.0), np.log(80.0), n_free))
    u = rng.uniform(0, KITTI_W, n_free)
    v = rng.uniform(0, KITTI_H, n_free)
    p_cam = np.stack([(u - cx) * z / f, (v - cy) * z / f, z], axis=1)
    p_veh_free = (p_cam - t_cv) @ R_cv  # R_cv^T (p - t)
    # ground points: vehicle frame of the newest keyframe, z = -height_over_ground, lateral +-25 m
    xg = rng.uniform(6.5, 45.0, n_ground)
    half = np.minimum(25.0, 0.8 * xg * (KITTI_W / 2) / f)
    yg = rng.uniform(-1, 1, n_ground) * half
    p_veh_ground = np.stack([xg, yg, np.full(n_ground, -HEIGHT_OVER_GROUND)], axis=1)
    p_veh = np.concatenate([p_veh_free, p_veh_ground], axis=0)
    is_ground = np.concatenate([np.zeros(n_free, np.uint8), np.ones(n_ground, np.uint8)])
    perm = rng.permutation(n_lm)
    p_veh, is_ground = p_veh[perm], is_ground[perm]
    gt_lm = p_veh @ R_ok[-1].T + p_ok[-1]  # origin frame

    # --- ground-truth projections, visibility
    R_co = np.einsum("ij,kjl->kil", R_cv, R_ko)  # camera <- origin
    t_co = np.einsum("ij,kj->ki", R_cv, t_ko) + t_cv
    pc = np.einsum("kij,nj->kni", R_co, gt_lm) + t_co[:, None, :]
    zc = pc[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        uu = f * pc[..., 0] / zc + cx
        vv = f * pc[..., 1] / zc + cy
    vis = (zc > 0.1) & (uu >= 0) & (uu < KITTI_W) & (vv >= 0) & (vv < KITTI_H)

    # --- measurements
    mu = uu + rng.normal(0, uv_sigma, uu.shape)
    mv = vv + rng.normal(0, uv_sigma, vv.shape)
    has_d = rng.uniform(size=uu.shape) < depth_prob
    md = zc + rng.normal(0, depth_sigma, zc.shape)
    gross = rng.uniform(size=uu.shape) < outlier_frac
    ang = rng.uniform(0, 2 * np.pi, uu.shape)
    mag = rng.uniform(5.0, 40.0, uu.shape)
    mu = np.where(gross, mu + mag * np.cos(ang), mu)
    mv = np.where(gross, mv + mag * np.sin(ang), mv)
    md = np.where(gross, md * rng.uniform(1.2, 2.0, uu.shape), md)
    md = np.where(has_d, md, -1.0)
    mu, mv, md = mu.astype(np.float32), mv.astype(np.float32), md.astype(np.float32)

    # --- initial poses: keyframe 0 exact and fixed, the others perturbed
    init_pose = gt_pose.copy()
    for k in range(1, n_kf):
        dR = rot_vec(rng.normal(0, np.deg2rad(rot_noise_deg), 3))
        dt_ = rng.normal(0, trans_noise, 3)
        init_pose[k] = Rt_to_pose(dR @ R_ko[k], dR @ t_ko[k] + dt_)
    Ri_ko, ti_ko = pose_to_Rt(init_pose)
    Ri_co = np.einsum("ij,kjl->kil", R_cv, Ri_ko)
    ti_co = np.einsum("ij,kj->ki", R_cv, ti_ko) + t_cv

    # --- landmark start values as push() produces them (bundle_adjuster_keyframes.cpp:289-382)
    lm_init, ok = init_landmarks(mu, mv, md, vis, Ri_co, ti_co, f, cx, cy)

    # --- cheirality rejection (default LandmarkSelector scheme): z >= 0 in every observing camera
    pci = np.einsum("kij,nj->kni", Ri_co, lm_init) + ti_co[:, None, :]
    cheiral = np.all((pci[..., 2] >= 0.0) | ~vis, axis=0)
    keep = ok & cheiral & (vis.sum(axis=0) > 0)
    idx = np.flatnonzero(keep)
    new_index = -np.ones(n_lm, np.int64)
    new_index[idx] = np.arange(idx.size)

    # --- flatten, keyframe-major like addKeyframeToProblem iterates (:505-507,:569)
    kk, nn = np.nonzero(vis[:, idx])
    lm_ids = idx[nn]
    weight = np.where(rng.uniform(size=n_lm) < shrub_frac, 0.9, 1.0)
    cams = cam[None, :]
    o_kf, o_lm, o_cam = kk.astype(np.int32), nn.astype(np.int32), np.zeros(kk.size, np.int32)
    o_u, o_v, o_d = mu[kk, lm_ids], mv[kk, lm_ids], md[kk, lm_ids]
    if stereo_baseline > 0.0:
        cam2 = cam.copy()
        cam2[7] = cam[7] - stereo_baseline  # camera <- vehicle translation of a camera displaced along its own +x
        t_cv2 = cam2[7:10]
        t_co2 = np.einsum("ij,kj->ki", R_cv, t_ko) + t_cv2
        pc2 = np.einsum("kij,nj->kni", R_co, gt_lm) + t_co2[:, None, :]
        z2 = pc2[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u2 = f * pc2[..., 0] / z2 + cx
            v2 = f * pc2[..., 1] / z2 + cy
        vis2 = (z2 > 0.1) & (u2 >= 0) & (u2 < KITTI_W) & (v2 >= 0) & (v2 < KITTI_H) & vis
        u2 = (u2 + rng.normal(0, uv_sigma, u2.shape)).astype(np.float32)
        v2 = (v2 + rng.normal(0, uv_sigma, v2.shape)).astype(np.float32)
        k2, n2 = np.nonzero(vis2[:, idx])
        l2 = idx[n2]
        cams = np.stack([cam, cam2])
        o_kf = np.concatenate([o_kf, k2.astype(np.int32)])
        o_lm = np.concatenate([o_lm, n2.astype(np.int32)])
        o_cam = np.concatenate([o_cam, np.ones(k2.size, np.int32)])
        o_u = np.concatenate([o_u, u2[k2, l2]])
        o_v = np.concatenate([o_v, v2[k2, l2]])
        o_d = np.concatenate([o_d, np.full(k2.size, -1.0, np.float32)])
    w = Window(
        kf_pose=init_pose,
        kf_plane_dir=np.tile(np.array([0.0, 0.0, 1.0]), (n_kf, 1)),
        kf_plane_dist=np.full(n_kf, HEIGHT_OVER_GROUND if with_ground_plane else -np.finfo(np.float64).max),
        kf_fixation=np.array(
            [_ffi.LIMO_FIX_POSE, _ffi.LIMO_FIX_SCALE] + [_ffi.LIMO_FIX_NONE] * (n_kf - 2), np.int32
        )[:n_kf],
        cam=cams,
        lm_pos=lm_init[idx],
        lm_weight=weight[idx],
        lm_is_ground=is_ground[idx] if with_ground_plane else np.zeros(idx.size, np.uint8),
        obs_kf=o_kf,
        obs_lm=o_lm,
        obs_cam=o_cam,
        obs_u=o_u,
        obs_v=o_v,
        obs_d=o_d,
        meta={"gt_pose": gt_pose, "gt_lm": gt_lm[idx], "seed": int(seed)},
    )
    return w


def init_landmarks(mu, mv, md, vis, R_co, t_co, f, cx, cy):
    """Vectorised BundleAdjusterKeyframes::push landmark creation over keyframes pushed in order.

    When keyframe k is pushed, an unknown landmark it measures is created from its depth (d >= 0,
    :332-355) or else by midpoint triangulation over all pushed keyframes that see it (>= 2 rays,
    :358-382, triangulator.hpp:51-75); otherwise it stays unknown until a later push.
    """
    K, N = vis.shape
    pos = np.zeros((N, 3))
    done = np.zeros(N, bool)
    # rays in origin frame and camera centres
    ray_c = np.stack([(mu.astype(np.float64) - cx) / f, (mv.astype(np.float64) - cy) / f, np.ones_like(mu, np.float64)], axis=-1)
    ray_c /= np.linalg.norm(ray_c, axis=-1, keepdims=True)
    R_oc = np.transpose(R_co, (0, 2, 1))
    c_o = -np.einsum("kij,kj->ki", R_oc, t_co)
    ray_o = np.einsum("kij,knj->kni", R_oc, ray_c)
    A = np.eye(3)[None, None] - ray_o[..., :, None] * ray_o[..., None, :]  # (K,N,3,3)
    b = np.einsum("knij,kj->kni", A, c_o)
    accA = np.zeros((N, 3, 3))
    accb = np.zeros((N, 3))
    nrays = np.zeros(N, np.int64)
    for k in range(K):
        seen = vis[k]
        accA[seen] += A[k][seen]
        accb[seen] += b[k][seen]
        nrays[seen] += 1
        # depth back-projection
        m = seen & ~done & (md[k] >= 0)
        if m.any():
            z = md[k][m].astype(np.float64)
            x = (mu[k][m].astype(np.float64) - cx) * z / f
            y = (mv[k][m].astype(np.float64) - cy) * z / f
            pc = np.stack([x, y, z], axis=1)
            pos[m] = (pc - t_co[k]) @ R_co[k]
            done[m] = True
        m = seen & ~done & (nrays >= 2)
        if m.any():
            pos[m] = np.einsum("nij,nj->ni", np.linalg.pinv(accA[m], rcond=3 * np.finfo(float).eps, hermitian=True), accb[m])
            done[m] = True
    return pos, done


def make_pose_only_case(seed):
    """New frame against fixed landmarks (adjustPoseOnly, bundle_adjuster_keyframes.cpp:769-904): the newest keyframe of a
    synthetic window as the frame to adjust, all landmarks at their ground-truth positions, the speed prior the shim
    would build from the two keyframes before it.  Returns (window, prior, ground-truth pose)."""
    w = make_window(seed, n_kf=4, n_lm=300, outlier_frac=0.02)
    k = w.n_kf - 1
    sel = w.obs_kf == k
    pw = Window(
        kf_pose=w.kf_pose[k : k + 1].copy(),
        kf_plane_dir=w.kf_plane_dir[k : k + 1],
        kf_plane_dist=w.kf_plane_dist[k : k + 1],
        kf_fixation=np.array([_ffi.LIMO_FIX_NONE], np.int32),
        cam=w.cam,
        lm_pos=w.meta["gt_lm"].copy(),
        lm_weight=w.lm_weight,
        lm_is_ground=w.lm_is_ground,
        obs_kf=np.zeros(sel.sum(), np.int32),
        obs_lm=w.obs_lm[sel],
        obs_cam=w.obs_cam[sel],
        obs_u=w.obs_u[sel],
        obs_v=w.obs_v[sel],
        obs_d=w.obs_d[sel],
    )
    gt = w.meta["gt_pose"][k]
    prior = _ffi.SpeedPrior()
    prior.speed_weight = 0.7
    prior.dt_cur = 0.4
    pb = w.meta["gt_pose"][k - 1]
    prior.pose_before[:] = pb.tolist()
    # velocity of the previous step expressed as the reference does: translation(pose_before * pose_before2^-1)/dt
    Rb, tb = pose_to_Rt(pb)
    Rbb, tbb = pose_to_Rt(w.meta["gt_pose"][k - 2])
    v = (tb - Rb @ Rbb.T @ tbb) / 0.4
    prior.vel_prev[:] = v.tolist()
    return pw, prior, gt


def make_batch(n, seed0=0, **kw):
    return [make_window(seed0 + i, **kw) for i in range(n)]


# named configurations of BASELINE.json
def config_c1(seed=1):
    """3 keyframes, 200 landmarks, reprojection only (no depth, no ground plane)."""
    return make_window(seed, n_kf=3, n_lm=200, depth_prob=0.0, ground_frac=0.0, with_ground_plane=False)


def config_c2(seed=2):
    """5 keyframes, ~2k landmarks, LiDAR depth + ground plane — the headline configuration."""
    return make_window(seed, n_kf=5, n_lm=2000)


def config_c4(seed=4):
    """10 keyframes, 8k landmarks."""
    return make_window(seed, n_kf=10, n_lm=8000)
