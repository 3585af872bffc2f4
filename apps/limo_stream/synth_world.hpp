// synth_world.hpp — synthetic "KITTI-00-shaped" drive for the streaming driver (BASELINE.json configs[4]): a static
// world (ground plane + boxes along the route), a 64-beam LiDAR sweep per frame (ray-cast, KITTI velodyne layout
// x,y,z,intensity; reference reader demo_keyframe_bundle_adjustment_meta/apps/main_program/utility.h:11-40) and tracked
// image features that sit ON the world's surfaces - they are drawn from earlier sweeps' hit points - with occlusion
// handled by ray-casting camera -> landmark.  Input synthesis only: nothing here is on the measured path.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

#include "../../limo_amd/kba/definitions.hpp"

namespace synth_world {

using keyframe_bundle_adjustment::EigenPose;
using keyframe_bundle_adjustment::Vector3d;
using matches_msg_types::FeaturePoint;
using matches_msg_types::Tracklet;
using matches_msg_types::Tracklets;

struct Box {
    double lo[3], hi[3];
};

// Three N(0,1) draws that depend only on `key` (splitmix64 + Box-Muller): a measurement is the same in every message.
inline void hash_normals(uint64_t key, double* out) {
    auto next = [&key]() {
        key += 0x9e3779b97f4a7c15ull;
        uint64_t z = key;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    };
    const double two_pi = 6.283185307179586;
    const double r0 = std::sqrt(-2.0 * std::log(next())), a0 = two_pi * next();
    const double r1 = std::sqrt(-2.0 * std::log(next())), a1 = two_pi * next();
    out[0] = r0 * std::cos(a0);
    out[1] = r0 * std::sin(a0);
    out[2] = r1 * std::cos(a1);
}

struct World {
    // geometry (vehicle frame: x forward, y left, z up; the vehicle origin drives 0.31 m above the ground, the LiDAR sits
    // 1.73 m above the ground, the camera looks forward)
    static constexpr double kHeightOverGround = 0.31, kLidarHeight = 1.73;
    double f = 718.856, cx = 607.1928, cy = 185.2157, W = 1241., H = 376.;
    EigenPose cam_veh, lidar_veh_inv /* vehicle <- lidar */, cam_lidar;
    std::vector<EigenPose> origin_veh;  // ground truth: origin <- vehicle_t
    std::vector<Box> boxes;
    int n_az = 2000;
    uint64_t seed;

    struct Landmark {
        Vector3d p;  // origin frame
        int born;    // frame whose sweep produced it
        bool ground;
    };
    std::vector<Landmark> lms;

    World(int n_frames, uint64_t seed_ = 7, double step = 0.55, double yaw_rate = 0.006) : seed(seed_) {
        const double R[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
        cam_veh = EigenPose::Identity();
        for (int i = 0; i < 9; ++i) cam_veh.R[i] = R[i];
        {
            const Vector3d c(1.08, 0., 1.35);  // camera position in the vehicle frame (KITTI static tf)
            const Vector3d t = cam_veh * Vector3d(-c[0], -c[1], -c[2]);
            cam_veh.t[0] = t[0] - cam_veh.t[0];
            cam_veh.t[1] = t[1] - cam_veh.t[1];
            cam_veh.t[2] = t[2] - cam_veh.t[2];
        }
        lidar_veh_inv = EigenPose::Identity();
        lidar_veh_inv.translate(Vector3d(0.8, 0., kLidarHeight - kHeightOverGround));  // lidar origin in the vehicle frame
        cam_lidar = cam_veh * lidar_veh_inv;
        std::mt19937_64 rng(seed);
        auto uni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
        EigenPose p = EigenPose::Identity();
        origin_veh.resize(n_frames);
        for (int t = 0; t < n_frames; ++t) {
            origin_veh[t] = p;
            p.translate(Vector3d(step, 0., 0.));
            p.rotate(yaw_rate, Vector3d(0., 0., 1.));
        }
        // boxes on both sides of the route: one every ~6 m of driven path, 4..22 m off the centre line
        const int n_boxes = std::max(8, (int)(n_frames * step / 6.));
        for (int b = 0; b < n_boxes; ++b) {
            const int at = (int)uni(0, n_frames - 1);
            const double side = uni(0, 1) < 0.5 ? -1. : 1.;
            const Vector3d c = origin_veh[at] * Vector3d(uni(-2., 2.), side * uni(4., 22.), 0.);
            const double sx = uni(0.5, 6.), sy = uni(0.5, 6.), sz = uni(1.2, 6.);
            Box bx;
            bx.lo[0] = c[0] - sx / 2;
            bx.hi[0] = c[0] + sx / 2;
            bx.lo[1] = c[1] - sy / 2;
            bx.hi[1] = c[1] + sy / 2;
            bx.lo[2] = -kHeightOverGround;
            bx.hi[2] = -kHeightOverGround + sz;
            boxes.push_back(bx);
        }
    }

    // first hit of the ray o + s d (s in (s_min, s_max)) with the ground or a box of `cand`; returns s or -1
    double cast(const Vector3d& o, const Vector3d& d, const std::vector<int>& cand, double s_min, double s_max, bool* on_ground = nullptr) const {
        double best = s_max;
        bool g = false;
        if (d[2] < -1e-9) {
            const double s = (-kHeightOverGround - o[2]) / d[2];
            if (s > s_min && s < best) {
                best = s;
                g = true;
            }
        }
        for (int bi : cand) {
            const Box& b = boxes[bi];
            double t0 = s_min, t1 = best;
            bool miss = false;
            for (int a = 0; a < 3 && !miss; ++a) {
                if (std::fabs(d[a]) < 1e-12) {
                    miss = o[a] < b.lo[a] || o[a] > b.hi[a];
                } else {
                    double ta = (b.lo[a] - o[a]) / d[a], tb = (b.hi[a] - o[a]) / d[a];
                    if (ta > tb) std::swap(ta, tb);
                    t0 = std::max(t0, ta);
                    t1 = std::min(t1, tb);
                    miss = t0 > t1;
                }
            }
            if (!miss && t0 > s_min && t0 < best) {
                best = t0;
                g = false;
            }
        }
        if (on_ground) *on_ground = g;
        return best < s_max ? best : -1.;
    }

    std::vector<int> boxes_near(const Vector3d& c, double radius) const {
        std::vector<int> out;
        for (size_t i = 0; i < boxes.size(); ++i) {
            const double dx = std::max({boxes[i].lo[0] - c[0], 0., c[0] - boxes[i].hi[0]});
            const double dy = std::max({boxes[i].lo[1] - c[1], 0., c[1] - boxes[i].hi[1]});
            if (dx * dx + dy * dy < radius * radius) out.push_back((int)i);
        }
        return out;
    }

    // 64-beam sweep of frame t in the LIDAR frame (x,y,z,intensity), + for each return whether it lies on the ground
    void sweep(int t, std::vector<float>& cloud, std::vector<uint8_t>& ground) const {
        const EigenPose origin_lidar = origin_veh[t] * lidar_veh_inv;
        const Vector3d o = origin_lidar.translation();
        const std::vector<int> cand = boxes_near(o, 85.);
        const int n_rays = 64 * n_az;
        std::vector<float> pts((size_t)n_rays * 4);
        std::vector<uint8_t> hit(n_rays, 0), gnd(n_rays, 0);
        auto work = [&](int r0, int r1) {
            for (int r = r0; r < r1; ++r) {
                const int beam = r / n_az, az = r % n_az;
                // HDL-64E S2 (the KITTI scanner): 32 lasers from +2 deg, 1/3 deg apart, then 32 from -8.83 deg, 1/2 deg apart
                const double el = (beam < 32 ? 2.0 - beam / 3.0 : -8.83 - (beam - 32) * 0.5) * M_PI / 180., a = -M_PI + 2 * M_PI * (az + 0.37) / n_az;
                const Vector3d dl(std::cos(el) * std::cos(a), std::cos(el) * std::sin(a), std::sin(el));
                Vector3d d;
                for (int i = 0; i < 3; ++i) d[i] = origin_lidar.R[3 * i] * dl[0] + origin_lidar.R[3 * i + 1] * dl[1] + origin_lidar.R[3 * i + 2] * dl[2];
                bool g;
                const double s = cast(o, d, cand, 0.5, 80., &g);
                if (s < 0) continue;
                double nz[3];
                hash_normals(seed * 7919ull + (uint64_t)t * 1000003ull + (uint64_t)r, nz);
                const double sr = s + 0.02 * nz[0];
                pts[4 * (size_t)r] = (float)(dl[0] * sr);
                pts[4 * (size_t)r + 1] = (float)(dl[1] * sr);
                pts[4 * (size_t)r + 2] = (float)(dl[2] * sr);
                pts[4 * (size_t)r + 3] = 0.5f;
                hit[r] = 1;
                gnd[r] = g;
            }
        };
        const int nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (int k = 0; k < nt; ++k) pool.emplace_back(work, n_rays * k / nt, n_rays * (k + 1) / nt);
        for (auto& th : pool) th.join();
        cloud.clear();
        ground.clear();
        for (int r = 0; r < n_rays; ++r)
            if (hit[r]) {
                for (int i = 0; i < 4; ++i) cloud.push_back(pts[4 * (size_t)r + i]);
                ground.push_back(gnd[r]);
            }
    }

    bool project(int t, const Vector3d& p_origin, double& u, double& v, double& z) const {
        const Vector3d pc = cam_veh * (origin_veh[t].inverse() * p_origin);
        z = pc[2];
        if (z < 1.5 || z > 60.) return false;
        u = f * pc[0] / z + cx;
        v = f * pc[1] / z + cy;
        return u >= 2 && u < W - 2 && v >= 2 && v < H - 2;
    }

    // visible = projects into the image and nothing lies between the camera and the point
    bool visible(int t, const Landmark& lm, const std::vector<int>& cand, double& u, double& v, double& z) const {
        if (!project(t, lm.p, u, v, z)) return false;
        const EigenPose origin_cam = origin_veh[t] * cam_veh.inverse();
        const Vector3d o = origin_cam.translation();
        Vector3d d(lm.p[0] - o[0], lm.p[1] - o[1], lm.p[2] - o[2]);
        const double len = d.norm();
        for (int i = 0; i < 3; ++i) d[i] /= len;
        return cast(o, d, cand, 0.2, len - 0.15) < 0;
    }

    // New landmarks from the sweep of frame t: hit points (origin frame) that project into the image, a fifth of them
    // on the ground.  Keeps about `target` landmarks in view.
    void spawn(int t, const std::vector<float>& cloud, const std::vector<uint8_t>& ground, int n_new, std::mt19937_64& rng) {
        const EigenPose origin_lidar = origin_veh[t] * lidar_veh_inv;
        std::vector<int> cg, co;
        for (size_t i = 0; i < ground.size(); ++i) {
            const Vector3d p = origin_lidar * Vector3d(cloud[4 * i], cloud[4 * i + 1], cloud[4 * i + 2]);
            double u, v, z;
            if (!project(t, p, u, v, z) || z < 4.) continue;
            (ground[i] ? cg : co).push_back((int)i);
        }
        auto take = [&](std::vector<int>& from, int n, bool g) {
            for (int k = 0; k < n && !from.empty(); ++k) {
                const size_t j = std::uniform_int_distribution<size_t>(0, from.size() - 1)(rng);
                const int i = from[j];
                from[j] = from.back();
                from.pop_back();
                lms.push_back({origin_lidar * Vector3d(cloud[4 * (size_t)i], cloud[4 * (size_t)i + 1], cloud[4 * (size_t)i + 2]), t, g});
            }
        };
        take(cg, n_new / 5, true);
        take(co, n_new - n_new / 5, false);
    }
};

}  // namespace synth_world
