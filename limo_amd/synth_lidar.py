"""Synthetic KITTI-shaped LiDAR sweeps and image features for the depth-assignment path.

A 64-beam spinning scanner (elevation +2 deg .. -24.8 deg, ~0.18 deg azimuth step => ~128 k returns per sweep in the
KITTI velodyne .bin layout x,y,z,intensity float32; reference reader:
demo_keyframe_bundle_adjustment_meta/apps/main_program/utility.h:11-40) is ray-cast against a ground plane and random
boxes; features are sampled at image locations where LiDAR returns project, as a feature tracker on real imagery
correlates with scene structure.  Pure numpy, seeded.
"""
import numpy as np

from .synth import KITTI_CX, KITTI_CY, KITTI_F, KITTI_H, KITTI_W, Rt_to_pose

LIDAR_HEIGHT = 1.73  # metres above ground (KITTI setup)


def kitti_T_cam_lidar():
    """camera <- lidar pose (w,x,y,z,t): lidar x fwd, y left, z up  ->  camera x right, y down, z fwd."""
    R = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    t = np.array([0.0, -0.08, -0.27])
    return Rt_to_pose(R, t)


def hdl64_elevations_deg():
    """Beam elevations of the Velodyne HDL-64E S2 that recorded KITTI: an upper block of 32 lasers from +2 deg down to
    -8.33 deg (1/3 deg apart) and a lower block of 32 from -8.83 deg to -24.33 deg (1/2 deg apart).  With KITTI's
    f = 719 px that is 4.2 px between scan lines over most of the image (upper block) and 6.3 px near its bottom: the
    9-pixel-high search window of the depth estimator (mono_lidar_fusion_parameters.yaml:17) then sees two scan lines
    almost everywhere in the upper block - with 64 evenly spread beams (5.4 px) half of the windows see only one line
    and no plane can be fitted."""
    return np.concatenate([2.0 - np.arange(32) / 3.0, -8.83 - np.arange(32) * 0.5])


def make_sweep(seed, n_az=2000, n_boxes=25, range_sigma=0.02, max_range=80.0):
    """Returns cloud [n,4] float32 (lidar frame).  Rays that hit nothing within max_range produce no return."""
    rng = np.random.default_rng(int(seed))
    elev = np.deg2rad(hdl64_elevations_deg())
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 2 * np.pi / n_az)
    E, A = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    t_hit = np.full(d.shape[0], np.inf)
    # ground plane z = -LIDAR_HEIGHT with a gentle tilt
    n = np.array([rng.normal(0, 0.01), rng.normal(0, 0.01), 1.0])
    n /= np.linalg.norm(n)
    dn = d @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = -LIDAR_HEIGHT / dn
    tg = np.where((dn < -1e-6) & (tg > 0), tg, np.inf)
    t_hit = np.minimum(t_hit, tg)
    # axis-aligned boxes (cars, walls, poles), mostly in front
    for _ in range(n_boxes):
        c = np.array([rng.uniform(5, 60), rng.uniform(-20, 20), 0.0])
        size = np.array([rng.uniform(0.3, 5.0), rng.uniform(0.3, 5.0), rng.uniform(1.0, 4.0)])
        lo = np.array([c[0] - size[0] / 2, c[1] - size[1] / 2, -LIDAR_HEIGHT])
        hi = np.array([c[0] + size[0] / 2, c[1] + size[1] / 2, -LIDAR_HEIGHT + size[2]])
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = lo / d
            t2 = hi / d
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmin > 0.5)
        t_hit = np.where(hit, np.minimum(t_hit, tmin), t_hit)
    ok = np.isfinite(t_hit) & (t_hit < max_range)
    r = t_hit[ok] + rng.normal(0, range_sigma, ok.sum())
    pts = d[ok] * r[:, None]
    inten = rng.uniform(0, 1, pts.shape[0])
    return np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)


def make_features(cloud, seed, n_feat=1500, ground_frac=0.2, jitter_px=2.0):
    """Feature pixels at projected-LiDAR-dense image locations; returns (uv float32 [n,2], is_ground uint8 [n], true depth)."""
    rng = np.random.default_rng(int(seed) + 7)
    from .synth import pose_to_Rt

    R, t = pose_to_Rt(kitti_T_cam_lidar())
    pc = cloud[:, :3].astype(np.float64) @ R.T + t
    z = pc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = KITTI_F * pc[:, 0] / z + KITTI_CX
        v = KITTI_F * pc[:, 1] / z + KITTI_CY
    vis = (z > 0.5) & (u >= 3) & (u < KITTI_W - 3) & (v >= 5) & (v < KITTI_H - 5)
    idx = np.flatnonzero(vis)
    on_ground = np.abs(cloud[idx, 2] + LIDAR_HEIGHT) < 0.15
    n_g = int(round(n_feat * ground_frac))
    gi = rng.choice(idx[on_ground], size=min(n_g, on_ground.sum()), replace=False)
    oi = rng.choice(idx[~on_ground], size=min(n_feat - gi.size, (~on_ground).sum()), replace=False)
    sel = np.concatenate([gi, oi])
    is_ground = np.concatenate([np.ones(gi.size, np.uint8), np.zeros(oi.size, np.uint8)])
    perm = rng.permutation(sel.size)
    sel, is_ground = sel[perm], is_ground[perm]
    uv = np.stack([u[sel], v[sel]], axis=1) + rng.uniform(-jitter_px, jitter_px, (sel.size, 2))
    return uv.astype(np.float32), is_ground, z[sel]


def make_frame(seed, n_feat=1500):
    cloud = make_sweep(seed)
    uv, is_ground, z_true = make_features(cloud, seed, n_feat)
    return {
        "cloud": cloud,
        "T_cam_lidar": kitti_T_cam_lidar(),
        "f": KITTI_F,
        "cx": KITTI_CX,
        "cy": KITTI_CY,
        "w": KITTI_W,
        "h": KITTI_H,
        "uv": uv,
        "is_ground": is_ground,
        "z_true": z_true,
    }


def visible_points(frame):
    """Number of sweep points in front of the camera that project into the image (the 20-byte records of SURVEY §8d D1)."""
    from .synth import pose_to_Rt

    R, t = pose_to_Rt(frame["T_cam_lidar"])
    pc = frame["cloud"][:, :3].astype(np.float64) @ R.T + t
    z = pc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = frame["f"] * pc[:, 0] / z + frame["cx"]
        v = frame["f"] * pc[:, 1] / z + frame["cy"]
    return int(((z > 0) & (u >= 0) & (u < frame["w"]) & (v >= 0) & (v < frame["h"])).sum())
