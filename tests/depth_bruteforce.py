"""A second, independently written LiDAR-depth assigner: brute-force numpy, no code shared with oracle/ or limo_amd/.

Why it exists.  The depth estimator LIMO uses (mono_lidar_depth) is not in the reference tree; what the tree pins is its
parameter file, demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml (cited as yaml:LINE).  The
oracle (oracle/depth_oracle.cpp) restates the estimator from that file, and the HIP kernels follow the oracle bit for
bit - so an error of interpretation in the oracle would be copied faithfully.  This file reads the SAME parameter file a
second time with different tools: dense numpy masks over all visible returns instead of cell lists, numpy's generator for
the RANSAC draws (other draws than the oracle's), an SVD for the least-squares planes, `itertools.combinations` for the
triangles.  tests/test_depth_bruteforce.py requires the two to make the same accept / reject decision for every feature
and to agree on the depth to 1e-5.

Reading of the parameter file (one line per key that takes part; defaults = the file's values):
  yaml:5    neighbor_search_mode 0: neighbours = projected returns inside a rectangle around the feature pixel
  yaml:14   pixelarea_search_witdh 6, yaml:17 pixelarea_search_height 9: full width / height, feature in the middle
            (yaml:21,24 offsets 0), borders included
  yaml:48   radiusSearch_count_min 3: fewer neighbours => outlier (-1)
  yaml:58   do_use_histogram_segmentation 1: histogram of the neighbours' depths, yaml:61 bin width 0.3 m, first bin
            starting at the nearest neighbour; "local maximum" = a bin holding more returns than the bin before it and at
            least as many as the bin after it, with >= yaml:63 min_pointcount 1 returns; of those the NEAREST one is kept
            (LIMO paper, README.md:45: foreground = "nearest significant bin"); no such bin => outlier
  yaml:171  do_use_triangle_size_maximation 1: local plane through the 3 returns of the bin that span the largest
            triangle (first triple in index order among equals); fewer than 3 returns => outlier
  yaml:173  do_check_triangleplanar_condition 1, yaml:176 threshold 0.1: sine of every inner angle ("crossnorm") >= 0.1
  yaml:178  viewray_plane_orthoganality_treshold 0.1: |cos(view ray, plane normal)| < 0.1 => outlier
  yaml:97   treshold_depth_enabled 1, mode 0 (yaml:99): depth outside (0, 100) m => outlier
  yaml:108  treshold_depth_local_enabled 1, mode 0, yaml:112 valuetype 1 (relative), yaml:114 value 0.5: depth outside
            [0.5 min z, 1.5 max z] of the returns the plane was built from => outlier
  yaml:168  do_use_cut_behind_camera 1: returns with z <= 0 in the camera frame do not exist
  yaml:128  do_use_ransac_plane 1 (features labelled ground): ground plane by RANSAC over the returns with lidar z in
            [yaml:131 -3.5, yaml:132 -1.0], inlier distance yaml:129 0.2 m, <= yaml:134 600 draws, stop probability
            yaml:136 0.99; yaml:138 refinement 1: least-squares plane over the band returns within yaml:140 10.2 m of it
  yaml:143  ransac_plane_point_distance_treshold 0.2: a ground feature's neighbours must lie within 0.2 m of that plane
  yaml:160  plane_estimator_use_mestimator 1: local patch = least squares weighted by the inverse distance to the ground
            plane, w = 1 / (|dist| + 1 cm) (the 1 cm keeps the weight finite); a patch tilted more than acos(0.9) against
            the ground plane (collinear returns of one scan line) or built from < 3 returns falls back to the ground plane
            itself, and then only the global depth gate applies
Output contract: FeaturePoint::d, float metres along the camera z axis, -1 = none
(matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26).
"""
import itertools

import numpy as np

P = dict(
    width=6, height=9, off_x=0, off_y=0, count_min=3,
    use_hist=True, bin_width=0.3, bin_min=1,
    gate=True, gate_min=0.0, gate_max=100.0,
    local_gate=True, local_relative=True, local_value=0.5,
    cut_behind=True, planar_check=True, planar_thr=0.1, ortho_thr=0.1,
    ransac=True, ransac_dist=0.2, band_lo=-3.5, band_hi=-1.0, ransac_iters=600, ransac_prob=0.99,
    refine=True, refine_thr=10.2, ground_pt_dist=0.2, mestimator=True,
)


def rotation(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def lsq_plane(pts, w=None):
    """(n, d) with n.p + d = 0 minimising the (weighted) squared distances: centroid + least singular direction."""
    w = np.ones(len(pts)) if w is None else np.asarray(w, float)
    c = (w[:, None] * pts).sum(0) / w.sum()
    _, _, vt = np.linalg.svd(np.sqrt(w)[:, None] * (pts - c), full_matrices=False)
    n = vt[-1]
    return n, -float(n @ c)


def ground_plane(frame, p=P, seed=12345):
    """Ground plane in the camera frame, oriented so that the camera is on its positive side; None if there is none."""
    cloud = np.asarray(frame["cloud"], np.float64)
    R, t = rotation(frame["T_cam_lidar"][:4]), np.asarray(frame["T_cam_lidar"][4:], float)
    band = cloud[(cloud[:, 2] >= p["band_lo"]) & (cloud[:, 2] <= p["band_hi"]), :3] @ R.T + t
    if len(band) < 3:
        return None
    rng = np.random.default_rng(seed)
    best, best_plane, need = 0, None, p["ransac_iters"]
    it = 0
    while it < min(need, p["ransac_iters"]):
        it += 1
        a, b, c = band[rng.choice(len(band), 3, replace=False)]
        n = np.cross(b - a, c - a)
        if np.linalg.norm(n) < 1e-9:
            continue
        n = n / np.linalg.norm(n)
        cnt = int((np.abs(band @ n - n @ a) < p["ransac_dist"]).sum())
        if cnt > best:
            best, best_plane = cnt, (n, -float(n @ a))
            w3 = (cnt / len(band)) ** 3
            need = 0 if w3 >= 1 else np.log(1 - p["ransac_prob"]) / np.log(1 - w3)
    if best < 3:
        return None
    n, d = best_plane
    if p["refine"]:
        near = np.abs(band @ n + d) < p["refine_thr"]
        if near.sum() >= 3:
            n, d = lsq_plane(band[near])
    return (n, d) if d >= 0 else (-n, -d)


def ray_depth(n, d, u, v, frame, p):
    r = np.array([(u - frame["cx"]) / frame["f"], (v - frame["cy"]) / frame["f"], 1.0])
    cosang = abs(n @ r) / np.linalg.norm(r)
    if cosang < p["ortho_thr"]:
        return None
    return -d / (n @ r)  # n.(s r) + d = 0, r_z = 1 => s is the depth


def estimate(frame, use_ground_labels=True, p=P):
    cloud = np.asarray(frame["cloud"], np.float64)
    R, t = rotation(frame["T_cam_lidar"][:4]), np.asarray(frame["T_cam_lidar"][4:], float)
    pc = cloud[:, :3] @ R.T + t
    front = pc[:, 2] > 0 if p["cut_behind"] else pc[:, 2] != 0
    pc = pc[front]
    u = frame["f"] * pc[:, 0] / pc[:, 2] + frame["cx"]
    v = frame["f"] * pc[:, 1] / pc[:, 2] + frame["cy"]
    inside = (u >= 0) & (u < frame["w"]) & (v >= 0) & (v < frame["h"])
    pc, u, v = pc[inside], u[inside], v[inside]
    uv = np.asarray(frame["uv"], np.float32).astype(np.float64)
    labels = np.asarray(frame["is_ground"], bool) if use_ground_labels else np.zeros(len(uv), bool)
    plane = ground_plane(frame, p) if (p["ransac"] and labels.any()) else None
    out = np.full(len(uv), -1.0, np.float32)
    for k, (fu, fv) in enumerate(uv):
        nb = pc[(np.abs(u - (fu + p["off_x"])) <= p["width"] / 2) & (np.abs(v - (fv + p["off_y"])) <= p["height"] / 2)]
        if len(nb) < p["count_min"]:
            continue
        if labels[k] and plane is not None:
            gn, gd = plane
            dist = nb @ gn + gd
            sel = np.abs(dist) < p["ground_pt_dist"]
            n_, d_, lo, hi = gn, gd, 0.0, np.inf
            if sel.sum() >= 3:
                w = 1.0 / (np.abs(dist[sel]) + 0.01) if p["mestimator"] else None
                ln, ld = lsq_plane(nb[sel], w)
                if abs(ln @ gn) >= 0.9:
                    n_, d_, lo, hi = ln, ld, nb[sel, 2].min(), nb[sel, 2].max()
            depth = ray_depth(n_, d_, fu, fv, frame, p)
        else:
            seg = nb
            if p["use_hist"]:
                z0 = nb[:, 2].min()
                b = np.floor((nb[:, 2] - z0) / p["bin_width"]).astype(int)
                cnt = np.bincount(b)
                pad = np.r_[0, cnt, 0]
                peaks = np.flatnonzero((cnt >= p["bin_min"]) & (cnt > pad[:-2]) & (cnt >= pad[2:]))
                if len(peaks) == 0:
                    continue
                seg = nb[b == peaks[0]]
            if len(seg) < 3:
                continue
            tri = np.array(list(itertools.combinations(range(len(seg)), 3)))
            A, B, C = seg[tri[:, 0]], seg[tri[:, 1]], seg[tri[:, 2]]
            area2 = (np.cross(B - A, C - A) ** 2).sum(1)
            A, B, C = (x[int(np.argmax(area2))] for x in (A, B, C))
            if p["planar_check"]:
                def sin_at(o, a, b):
                    e1, e2 = a - o, b - o
                    l = np.linalg.norm(e1) * np.linalg.norm(e2)
                    return np.linalg.norm(np.cross(e1, e2)) / l if l > 0 else 0.0
                if min(sin_at(A, B, C), sin_at(B, A, C), sin_at(C, A, B)) < p["planar_thr"]:
                    continue
            n_ = np.cross(B - A, C - A)
            if np.linalg.norm(n_) == 0:
                continue
            n_ = n_ / np.linalg.norm(n_)
            depth = ray_depth(n_, -float(n_ @ A), fu, fv, frame, p)
            lo, hi = seg[:, 2].min(), seg[:, 2].max()
        if depth is None:
            continue
        if p["gate"] and not (p["gate_min"] < depth < p["gate_max"]):
            continue
        if p["local_gate"]:
            a, b = (lo * (1 - p["local_value"]), hi * (1 + p["local_value"])) if p["local_relative"] else (lo - p["local_value"], hi + p["local_value"])
            if not (a <= depth <= b):
                continue
        out[k] = depth
    return out
