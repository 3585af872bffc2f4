// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// depth_oracle.cpp — CPU restatement of the LiDAR->feature depth assignment (SURVEY §8a rows D1–D6).
//
// PARITY UNPINNED: the implementation LIMO uses lives in the un-vendored repository johannes-graeter/mono_lidar_depth
// (packages monolidar_fusion / tracklets_depth, cloned at unpinned HEAD by install_repos.sh:9 and
// docker/src/Dockerfile:71-72); nothing of it is under /root/reference and no reference test touches it.  What the
// tree pins is the parameter file demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml (cited
// below as yaml:LINE), the output contract FeaturePoint::d (matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26,
// float metres along camera z, -1 = none) and the method description of the LIMO paper cited at README.md:45
// ("nearest significant histogram bin", "plane through the three points spanning the largest triangle").
// Where the parameter file leaves a choice open, the choice made here is spelled out next to the code; the HIP
// implementation (limo_amd/csrc/depth.hip) makes the same choices and is tested against this file BIT FOR BIT
// (accept / reject decisions and depths).  What makes that possible is part of this file's contract:
//   * it is compiled with -ffp-contract=off (oracle/Makefile): every + - * / sqrt below is one IEEE-754 operation in
//     the order written;
//   * the one long floating-point sum of the path, the moments of the ground-plane refinement, is specified in FIXED
//     POINT (int64; units 2^-30 m for first, 2^-20 m^2 for second moments, round-to-nearest-even per term): integer
//     sums do not depend on the order of summation, so a parallel implementation can form the same numbers.  The
//     rounding per term (<= 2^-21 m^2) is 10 orders of magnitude below the scatter of a ground band.
// An implementation written independently of this file (tests/depth_bruteforce.py: numpy, from the parameter file only)
// cross-checks the choices on analytic and synthetic scenes.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/limo_hip.h"

namespace {

struct Vis {  // a lidar return that projects into the image
    int idx;
    double u, v, x, y, z;  // pixel, camera-frame position
};

struct Plane4 {
    double n[3], d;  // n.p + d = 0
    bool ok;
};

void quat_to_R(const double* q, double* R) {  // same polynomial form as the BA path
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - w * z);
    R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);
    R[7] = 2 * (y * z + w * x);
    R[8] = 1 - 2 * (x * x + y * y);
}

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 matrix (cyclic Jacobi)
void smallest_eigvec(const double C[6], double* n) {  // C = xx xy xz yy yz zz (upper triangle, mirrored)
    double a[3][3] = {{C[0], C[1], C[2]}, {C[1], C[3], C[4]}, {C[2], C[4], C[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * diag || off < 1e-300) break;  // converged to far below the rounding of the entries
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                if (a[i][j] == 0.0) continue;
                const double tau = (a[j][j] - a[i][i]) / (2.0 * a[i][j]);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    const double x = a[k][i], y = a[k][j];
                    a[k][i] = cs * x - sn * y;
                    a[k][j] = sn * x + cs * y;
                }
                for (int k = 0; k < 3; ++k) {
                    const double x = a[i][k], y = a[j][k];
                    a[i][k] = cs * x - sn * y;
                    a[j][k] = sn * x + cs * y;
                }
                for (int k = 0; k < 3; ++k) {
                    const double x = V[k][i], y = V[k][j];
                    V[k][i] = cs * x - sn * y;
                    V[k][j] = sn * x + cs * y;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    double nn = std::sqrt(V[0][m] * V[0][m] + V[1][m] * V[1][m] + V[2][m] * V[2][m]);
    for (int k = 0; k < 3; ++k) n[k] = V[k][m] / nn;
}

// weighted total-least-squares plane through points (centroid + smallest eigenvector of the weighted scatter)
Plane4 fit_plane(const std::vector<const double*>& pts, const std::vector<double>& w) {
    Plane4 P;
    P.ok = false;
    if (pts.size() < 3) return P;
    double sw = 0, c[3] = {0, 0, 0};
    for (size_t i = 0; i < pts.size(); ++i) {
        sw += w[i];
        for (int k = 0; k < 3; ++k) c[k] += w[i] * pts[i][k];
    }
    for (int k = 0; k < 3; ++k) c[k] /= sw;
    double C[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < pts.size(); ++i) {
        const double d[3] = {pts[i][0] - c[0], pts[i][1] - c[1], pts[i][2] - c[2]};
        C[0] += w[i] * d[0] * d[0];
        C[1] += w[i] * d[0] * d[1];
        C[2] += w[i] * d[0] * d[2];
        C[3] += w[i] * d[1] * d[1];
        C[4] += w[i] * d[1] * d[2];
        C[5] += w[i] * d[2] * d[2];
    }
    smallest_eigvec(C, P.n);
    P.d = -(P.n[0] * c[0] + P.n[1] * c[1] + P.n[2] * c[2]);
    P.ok = true;
    return P;
}

// depth along the viewing ray of pixel (u,v) where it meets the plane; false if (nearly) parallel
bool ray_plane_depth(const Plane4& P, double u, double v, double f, double cx, double cy, double ortho_thr, double* depth) {
    const double r[3] = {(u - cx) / f, (v - cy) / f, 1.0};
    const double rn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double nr = P.n[0] * r[0] + P.n[1] * r[1] + P.n[2] * r[2];
    if (std::fabs(nr / rn) < ortho_thr) return false;  // yaml:178 viewray_plane_orthoganality_treshold
    *depth = -P.d / nr;                                // s with n.(s r) + d = 0; r_z == 1 => depth = s
    return true;
}

}  // namespace

extern "C" {

void oracle_depth_default_params(limo_depth_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->pixelarea_search_width = 6;        // yaml:14
    p->pixelarea_search_height = 9;       // yaml:17
    p->neighbors_count_min = 3;           // yaml:48
    p->do_use_histogram_segmentation = 1; // yaml:58
    p->histogram_segmentation_bin_width = 0.3;  // yaml:61
    p->histogram_segmentation_min_pointcount = 1;  // yaml:63
    p->treshold_depth_enabled = 1;        // yaml:97
    p->treshold_depth_max = 100.0;        // yaml:101
    p->treshold_depth_min = 0.0;          // yaml:103
    p->treshold_depth_local_enabled = 1;  // yaml:108
    p->treshold_depth_local_valuetype = 1;  // yaml:112
    p->treshold_depth_local_value = 0.5;  // yaml:114
    p->do_use_cut_behind_camera = 1;      // yaml:168
    p->do_use_triangle_size_maximation = 1;  // yaml:171
    p->do_check_triangleplanar_condition = 1;  // yaml:173
    p->triangleplanar_crossnorm_treshold = 0.1;  // yaml:176
    p->viewray_plane_orthoganality_treshold = 0.1;  // yaml:178
    p->do_use_ransac_plane = 1;           // yaml:128
    p->ransac_plane_distance_treshold = 0.2;  // yaml:129
    p->ransac_plane_min_z = -3.5;         // yaml:131
    p->ransac_plane_max_z = -1.0;         // yaml:132
    p->ransac_plane_max_iterations = 600; // yaml:134
    p->ransac_plane_probability = 0.99;   // yaml:136
    p->ransac_plane_use_refinement = 1;   // yaml:138
    p->ransac_plane_refinement_treshold = 10.2;  // yaml:140
    p->ransac_plane_point_distance_treshold = 0.2;  // yaml:143
    p->plane_estimator_use_mestimator = 1;  // yaml:160
    p->ransac_seed = 1;
}

// Ground plane in the CAMERA frame from the lidar sweep: RANSAC over the returns with lidar z in [min_z, max_z]
// (yaml:128-136), adaptive iteration count from `probability`, then least-squares refinement over the band points
// within `refinement_treshold` of the RANSAC plane (yaml:138-140).  plane4 = (n, d) with n.p + d = 0, n pointing
// towards the camera (n.y < 0 in a camera frame whose y axis points down).  Returns the number of RANSAC inliers, 0
// if no plane.
int oracle_ground_plane(const float* cloud, size_t n_pts, const double* T_cam_lidar, const limo_depth_params* p,
                        double* plane4) {
    double R[9];
    quat_to_R(T_cam_lidar, R);
    const double* t = T_cam_lidar + 4;
    std::vector<std::array<double, 3>> band;
    for (size_t i = 0; i < n_pts; ++i) {
        const double z = cloud[4 * i + 2];
        if (!(z >= p->ransac_plane_min_z && z <= p->ransac_plane_max_z)) continue;  // NaN returns are dropped
        const double x = cloud[4 * i], y = cloud[4 * i + 1];
        band.push_back({R[0] * x + R[1] * y + R[2] * z + t[0], R[3] * x + R[4] * y + R[5] * z + t[1],
                        R[6] * x + R[7] * y + R[8] * z + t[2]});
    }
    const size_t nb = band.size();
    if (nb < 3) return 0;
    int best = 0;
    double bn[3] = {0, 0, 0}, bd = 0;
    double k_needed = p->ransac_plane_max_iterations;
    for (int it = 0; it < p->ransac_plane_max_iterations; ++it) {
        if (it >= k_needed) break;  // adaptive stop (standard RANSAC: k = log(1-p)/log(1-w^3))
        const uint64_t h = splitmix64(p->ransac_seed * 0x100000001B3ull + (uint64_t)it);
        const size_t i0 = splitmix64(h) % nb, i1 = splitmix64(h + 1) % nb, i2 = splitmix64(h + 2) % nb;
        if (i0 == i1 || i0 == i2 || i1 == i2) continue;
        const double* a = band[i0].data();
        const double* b = band[i1].data();
        const double* c = band[i2].data();
        const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nn > 1e-9)) continue;
        for (int k = 0; k < 3; ++k) n[k] /= nn;
        const double d = -(n[0] * a[0] + n[1] * a[1] + n[2] * a[2]);
        int cnt = 0;
        for (size_t q = 0; q < nb; ++q)
            if (std::fabs(n[0] * band[q][0] + n[1] * band[q][1] + n[2] * band[q][2] + d) < p->ransac_plane_distance_treshold) ++cnt;
        if (cnt > best) {
            best = cnt;
            for (int k = 0; k < 3; ++k) bn[k] = n[k];
            bd = d;
            const double w = (double)cnt / (double)nb;
            const double denom = std::log(std::max(1e-300, 1.0 - w * w * w));
            k_needed = denom < 0 ? std::log(1.0 - p->ransac_plane_probability) / denom : 0.0;
        }
    }
    if (best < 3) return 0;
    if (p->ransac_plane_use_refinement) {
        // centroid and scatter matrix from the moments of e = p - a around the point a = -d n of the RANSAC plane (the
        // returns lie around it, so nothing cancels), accumulated in fixed point (see the header): first moments in units
        // of 2^-30 m, second moments in units of 2^-20 m^2, each term rounded to nearest-even; a return further than
        // 1024 m from a in any coordinate does not take part.
        const double kScale1 = 1073741824.0, kScale2 = 1048576.0, kRange = 1024.0;
        const double a[3] = {-bd * bn[0], -bd * bn[1], -bd * bn[2]};
        long long mom[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t q = 0; q < nb; ++q) {
            if (!(std::fabs(bn[0] * band[q][0] + bn[1] * band[q][1] + bn[2] * band[q][2] + bd) < p->ransac_plane_refinement_treshold)) continue;
            const double e[3] = {band[q][0] - a[0], band[q][1] - a[1], band[q][2] - a[2]};
            if (!(std::fabs(e[0]) < kRange && std::fabs(e[1]) < kRange && std::fabs(e[2]) < kRange)) continue;
            mom[0] += 1;
            mom[1] += std::llrint(e[0] * kScale1);
            mom[2] += std::llrint(e[1] * kScale1);
            mom[3] += std::llrint(e[2] * kScale1);
            mom[4] += std::llrint(e[0] * e[0] * kScale2);
            mom[5] += std::llrint(e[0] * e[1] * kScale2);
            mom[6] += std::llrint(e[0] * e[2] * kScale2);
            mom[7] += std::llrint(e[1] * e[1] * kScale2);
            mom[8] += std::llrint(e[1] * e[2] * kScale2);
            mom[9] += std::llrint(e[2] * e[2] * kScale2);
        }
        const double m0 = (double)mom[0];
        if (m0 >= 3.0) {
            const double c[3] = {((double)mom[1] / kScale1) / m0, ((double)mom[2] / kScale1) / m0, ((double)mom[3] / kScale1) / m0};  // centroid - a
            const double S[6] = {(double)mom[4] / kScale2 - m0 * c[0] * c[0], (double)mom[5] / kScale2 - m0 * c[0] * c[1],
                                 (double)mom[6] / kScale2 - m0 * c[0] * c[2], (double)mom[7] / kScale2 - m0 * c[1] * c[1],
                                 (double)mom[8] / kScale2 - m0 * c[1] * c[2], (double)mom[9] / kScale2 - m0 * c[2] * c[2]};
            smallest_eigvec(S, bn);
            bd = -(bn[0] * (a[0] + c[0]) + bn[1] * (a[1] + c[1]) + bn[2] * (a[2] + c[2]));
        }
    }
    if (bd < 0) {  // orient: the camera (origin) is on the positive side
        for (int k = 0; k < 3; ++k) bn[k] = -bn[k];
        bd = -bd;
    }
    plane4[0] = bn[0];
    plane4[1] = bn[1];
    plane4[2] = bn[2];
    plane4[3] = bd;
    return best;
}

int oracle_depth_estimate(const float* cloud, size_t n_pts, const double* T_cam_lidar, double f, double cx, double cy,
                          int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground,
                          const limo_depth_params* p, float* depth_out) {
    double R[9];
    quat_to_R(T_cam_lidar, R);
    const double* t = T_cam_lidar + 4;
    // ---- D1: lidar -> camera, cut behind camera (yaml:168), pinhole projection, in-image test
    std::vector<Vis> vis;
    for (size_t i = 0; i < n_pts; ++i) {
        const double x = cloud[4 * i], y = cloud[4 * i + 1], z = cloud[4 * i + 2];
        Vis q;
        q.idx = (int)i;
        q.x = R[0] * x + R[1] * y + R[2] * z + t[0];
        q.y = R[3] * x + R[4] * y + R[5] * z + t[1];
        q.z = R[6] * x + R[7] * y + R[8] * z + t[2];
        if (p->do_use_cut_behind_camera && !(q.z > 0.0)) continue;
        if (q.z == 0.0) continue;
        q.u = f * q.x / q.z + cx;
        q.v = f * q.y / q.z + cy;
        if (!(q.u >= 0.0 && q.u < (double)img_w && q.v >= 0.0 && q.v < (double)img_h)) continue;
        vis.push_back(q);
    }
    // ---- D6a: ground plane (only if any feature is labelled ground)
    Plane4 ground;
    ground.ok = false;
    bool any_ground = false;
    if (feat_is_ground)
        for (size_t k = 0; k < n_feat; ++k) any_ground = any_ground || feat_is_ground[k];
    if (any_ground && p->do_use_ransac_plane) {
        double pl[4];
        if (oracle_ground_plane(cloud, n_pts, T_cam_lidar, p, pl) > 0) {
            ground.ok = true;
            ground.n[0] = pl[0];
            ground.n[1] = pl[1];
            ground.n[2] = pl[2];
            ground.d = pl[3];
        }
    }
    const double hw = 0.5 * p->pixelarea_search_width, hh = 0.5 * p->pixelarea_search_height;
    // per-gate rejection counts (ORACLE_DEPTH_STATS=1 prints them: which gate starves a workload of depths)
    enum { G_NEIGHBOURS, G_HISTOGRAM, G_SEGMENT3, G_PLANAR, G_PARALLEL, G_GLOBAL, G_LOCAL, G_OK, G_N };
    long gate[G_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t k = 0; k < n_feat; ++k) {
        depth_out[k] = -1.0f;
        const double fu = feat_uv[2 * k], fv = feat_uv[2 * k + 1];
        // ---- D2: neighbours = visible returns inside the pixel rectangle centred on the feature (yaml:5,14,17,21,24);
        //      bounds inclusive; list in lidar-point order
        std::vector<const Vis*> nb;
        for (const Vis& q : vis)
            if (std::fabs(q.u - (fu + p->pixelarea_search_offset_x)) <= hw && std::fabs(q.v - (fv + p->pixelarea_search_offset_y)) <= hh)
                nb.push_back(&q);
        if ((int)nb.size() < p->neighbors_count_min) {  // yaml:48
            gate[G_NEIGHBOURS]++;
            continue;
        }
        double depth = -1.0;
        double zlo, zhi;
        if (feat_is_ground && feat_is_ground[k] && ground.ok) {
            // ---- D6b: ground feature: neighbours within point_distance_treshold of the ground plane (yaml:143), local
            //      patch by inverse-distance weighted least squares (yaml:160); fewer than 3 -> the global plane
            std::vector<const double*> pts;
            std::vector<double> w;
            zlo = std::numeric_limits<double>::max();
            zhi = -zlo;
            for (const Vis* q : nb) {
                const double dist = ground.n[0] * q->x + ground.n[1] * q->y + ground.n[2] * q->z + ground.d;
                if (std::fabs(dist) < p->ransac_plane_point_distance_treshold) {
                    pts.push_back(&q->x);
                    w.push_back(p->plane_estimator_use_mestimator ? 1.0 / (std::fabs(dist) + 0.01) : 1.0);
                    zlo = std::min(zlo, q->z);
                    zhi = std::max(zhi, q->z);
                }
            }
            Plane4 P = fit_plane(pts, w);
            // a patch from (nearly) collinear returns of one scan line is ill-conditioned: accept the local patch only
            // if it is roughly parallel to the sweep's ground plane (|n_local . n_ground| >= 0.9), else use that plane
            if (P.ok && std::fabs(P.n[0] * ground.n[0] + P.n[1] * ground.n[1] + P.n[2] * ground.n[2]) < 0.9) P.ok = false;
            if (!P.ok) {
                P = ground;
                zlo = 0.0;
                zhi = std::numeric_limits<double>::max();
            }
            if (!ray_plane_depth(P, fu, fv, f, cx, cy, p->viewray_plane_orthoganality_treshold, &depth)) {
                gate[G_PARALLEL]++;
                continue;
            }
        } else {
            // ---- D3: histogram segmentation by camera depth, bin width yaml:61, from the nearest neighbour's depth;
            //      the NEAREST bin that is a local maximum with >= min_pointcount points is kept (LIMO paper: "nearest
            //      significant bin"); none -> reject (yaml:58-63)
            std::vector<const Vis*> seg;
            if (p->do_use_histogram_segmentation) {
                double zmin = std::numeric_limits<double>::max(), zmax = -zmin;
                for (const Vis* q : nb) {
                    zmin = std::min(zmin, q->z);
                    zmax = std::max(zmax, q->z);
                }
                const double bw = p->histogram_segmentation_bin_width;
                const int nbins = (int)std::floor((zmax - zmin) / bw) + 1;
                std::vector<int> cnt(nbins, 0);
                for (const Vis* q : nb) cnt[std::min(nbins - 1, (int)std::floor((q->z - zmin) / bw))]++;
                int pick = -1;
                for (int b = 0; b < nbins && pick < 0; ++b) {
                    const int prev = b > 0 ? cnt[b - 1] : 0, next = b + 1 < nbins ? cnt[b + 1] : 0;
                    if (cnt[b] >= p->histogram_segmentation_min_pointcount && cnt[b] > prev && cnt[b] >= next) pick = b;
                }
                if (pick < 0) {
                    gate[G_HISTOGRAM]++;
                    continue;
                }
                for (const Vis* q : nb)
                    if (std::min(nbins - 1, (int)std::floor((q->z - zmin) / bw)) == pick) seg.push_back(q);
            } else {
                seg = nb;
            }
            if (seg.size() < 3) {
                gate[G_SEGMENT3]++;
                continue;
            }
            zlo = std::numeric_limits<double>::max();
            zhi = -zlo;
            for (const Vis* q : seg) {
                zlo = std::min(zlo, q->z);
                zhi = std::max(zhi, q->z);
            }
            // ---- D4: plane through the three points spanning the largest triangle (yaml:171); ties -> first in
            //      lexicographic (i<j<k) order
            double best = -1.0;
            int bi = -1, bj = -1, bk = -1;
            const int n = (int)seg.size();
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j)
                    for (int l = j + 1; l < n; ++l) {
                        const double e1[3] = {seg[j]->x - seg[i]->x, seg[j]->y - seg[i]->y, seg[j]->z - seg[i]->z};
                        const double e2[3] = {seg[l]->x - seg[i]->x, seg[l]->y - seg[i]->y, seg[l]->z - seg[i]->z};
                        const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                        const double a2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
                        if (a2 > best) {
                            best = a2;
                            bi = i;
                            bj = j;
                            bk = l;
                        }
                    }
            if (bi < 0) continue;
            const double* A = &seg[bi]->x;
            const double* B = &seg[bj]->x;
            const double* Cc = &seg[bk]->x;
            // planarity gate (yaml:173-176): the sine of every inner angle of the triangle must reach the threshold
            auto sin_at = [](const double* o, const double* a, const double* b) {
                const double e1[3] = {a[0] - o[0], a[1] - o[1], a[2] - o[2]}, e2[3] = {b[0] - o[0], b[1] - o[1], b[2] - o[2]};
                const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), n2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                if (!(n1 > 0.0) || !(n2 > 0.0)) return 0.0;
                return std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / (n1 * n2);
            };
            if (p->do_check_triangleplanar_condition) {
                const double s = std::min(sin_at(A, B, Cc), std::min(sin_at(B, A, Cc), sin_at(Cc, A, B)));
                if (s < p->triangleplanar_crossnorm_treshold) {
                    gate[G_PLANAR]++;
                    continue;
                }
            }
            Plane4 P;
            const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {Cc[0] - A[0], Cc[1] - A[1], Cc[2] - A[2]};
            P.n[0] = e1[1] * e2[2] - e1[2] * e2[1];
            P.n[1] = e1[2] * e2[0] - e1[0] * e2[2];
            P.n[2] = e1[0] * e2[1] - e1[1] * e2[0];
            const double nn = std::sqrt(P.n[0] * P.n[0] + P.n[1] * P.n[1] + P.n[2] * P.n[2]);
            if (!(nn > 0.0)) continue;
            for (int q = 0; q < 3; ++q) P.n[q] /= nn;
            P.d = -(P.n[0] * A[0] + P.n[1] * A[1] + P.n[2] * A[2]);
            P.ok = true;
            if (!ray_plane_depth(P, fu, fv, f, cx, cy, p->viewray_plane_orthoganality_treshold, &depth)) {
                gate[G_PARALLEL]++;
                continue;
            }
        }
        // ---- D5: global gate (yaml:97-103, mode 0: reject) and local gate relative to the depth range of the points the
        //      patch was built from (yaml:108-114)
        if (p->treshold_depth_enabled && !(depth > p->treshold_depth_min && depth < p->treshold_depth_max)) {
            gate[G_GLOBAL]++;
            continue;
        }
        if (p->treshold_depth_local_enabled) {
            const double v = p->treshold_depth_local_value;
            const double lo = p->treshold_depth_local_valuetype ? zlo * (1.0 - v) : zlo - v;
            const double hi = p->treshold_depth_local_valuetype ? zhi * (1.0 + v) : zhi + v;
            if (!(depth >= lo && depth <= hi)) {
                gate[G_LOCAL]++;
                continue;
            }
        }
        gate[G_OK]++;
        depth_out[k] = (float)depth;
    }
    if (std::getenv("ORACLE_DEPTH_STATS"))
        std::fprintf(stderr, "[depth oracle] %zu features, %zu visible points: < 3 neighbours %ld, no histogram bin %ld, segment < 3 %ld, triangle not planar enough %ld, ray || plane %ld, global gate %ld, local gate %ld, accepted %ld\n",
                     n_feat, vis.size(), gate[G_NEIGHBOURS], gate[G_HISTOGRAM], gate[G_SEGMENT3], gate[G_PLANAR], gate[G_PARALLEL], gate[G_GLOBAL], gate[G_LOCAL], gate[G_OK]);
    return LIMO_OK;
}

}  // extern "C"
