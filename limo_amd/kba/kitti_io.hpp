// kitti_io.hpp — the data formats either side of the hot path:
//   * KITTI velodyne scans (`velodyne/NNNNNN.bin`, float32 x,y,z,intensity back to back) -> the flat [n*4] buffer that
//     limo_depth_estimate takes as it is.  Replaces read_lidar_data of the reference's demo application
//     (demo_keyframe_bundle_adjustment_meta/apps/main_program/utility.h:11-40), which unpacks the same file into a
//     pcl::PointCloud + an Eigen vector + an intensity vector; here nothing is unpacked - the file layout IS the device
//     layout (16-byte records, one coalesced load per return in k_project).
//   * KITTI odometry pose files (12 numbers per line = the top 3 rows of camera_0 <- camera_k, row-major), written by the
//     node at mono_lidar.cpp:281-294 and read by the KITTI devkit.
//   * trajectory error measures on such pose lists: absolute trajectory error (positions, no alignment: both start at
//     the identity) and the devkit-style relative errors over fixed path lengths.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "definitions.hpp"

namespace keyframe_bundle_adjustment {
namespace kitti_io {

// false: file missing or its size is not a multiple of 16 bytes.  xyzi receives 4 floats per return.
inline bool readVelodyneBin(const std::string& path, std::vector<float>& xyzi) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    bool ok = bytes >= 0 && bytes % 16 == 0;
    if (ok) {
        xyzi.resize((size_t)bytes / sizeof(float));
        ok = xyzi.empty() || std::fread(xyzi.data(), sizeof(float), xyzi.size(), f) == xyzi.size();
    }
    std::fclose(f);
    return ok;
}
inline bool writeVelodyneBin(const std::string& path, const float* xyzi, size_t n_pts) {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = n_pts == 0 || std::fwrite(xyzi, 4 * sizeof(float), n_pts, f) == n_pts;
    return std::fclose(f) == 0 && ok;
}
// "%s/%06d.bin": the scan of frame k inside a KITTI sequence's velodyne directory
inline std::string velodynePath(const std::string& dir, int frame) {
    char name[32];
    std::snprintf(name, sizeof(name), "/%06d.bin", frame);
    return dir + name;
}

inline void writePoseRow(std::ostream& os, const EigenPose& m) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g", m.R[0], m.R[1], m.R[2], m.t[0], m.R[3],
                  m.R[4], m.R[5], m.t[1], m.R[6], m.R[7], m.R[8], m.t[2]);
    os << buf << "\n";
}
inline bool readPoses(const std::string& path, std::vector<EigenPose>& out) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
        std::istringstream ss(line);
        double v[12];
        for (double& x : v)
            if (!(ss >> x)) return false;
        EigenPose p;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) p.R[3 * r + c] = v[4 * r + c];
            p.t[r] = v[4 * r + 3];
        }
        out.push_back(p);
    }
    return true;
}

struct TrajectoryError {
    double ate_rmse = 0., ate_max = 0.;     // metres, positions of est vs gt frame by frame
    double rel_trans = 0., rel_rot = 0.;    // devkit style: mean translational error (fraction of the path) and rotational
    int rel_samples = 0;                    // error (rad per metre) over sub-paths of the listed lengths
    double path_length = 0.;
};
// Poses are camera_0 <- camera_k (what the pose files hold).  Sub-paths start every `step` frames and span each of
// `lengths` metres of ground-truth path (KITTI devkit: 100..800 m every 10 frames); the error of a sub-path is the
// motion est^-1 * gt of its end relative to its start.
inline TrajectoryError evaluateTrajectory(const std::vector<EigenPose>& gt, const std::vector<EigenPose>& est,
                                          const std::vector<double>& lengths = {100, 200, 300, 400, 500, 600, 700, 800}, int step = 10) {
    TrajectoryError e;
    const size_t n = std::min(gt.size(), est.size());
    if (n == 0) return e;
    std::vector<double> dist(n, 0.);
    double se = 0.;
    for (size_t k = 0; k < n; ++k) {
        const Vector3d d = est[k].translation() - gt[k].translation();
        se += d.norm() * d.norm();
        e.ate_max = std::max(e.ate_max, d.norm());
        if (k) dist[k] = dist[k - 1] + (gt[k].translation() - gt[k - 1].translation()).norm();
    }
    e.ate_rmse = std::sqrt(se / (double)n);
    e.path_length = dist[n - 1];
    for (size_t first = 0; first < n; first += (size_t)std::max(1, step))
        for (double len : lengths) {
            size_t last = first;
            while (last < n && dist[last] < dist[first] + len) ++last;
            if (last >= n) continue;
            const EigenPose dg = gt[first].inverse() * gt[last], de = est[first].inverse() * est[last];
            const EigenPose err = de.inverse() * dg;
            const double tr = err.R[0] + err.R[4] + err.R[8];
            e.rel_rot += std::acos(std::max(-1., std::min(1., 0.5 * (tr - 1.)))) / len;
            e.rel_trans += err.translation().norm() / len;
            ++e.rel_samples;
        }
    if (e.rel_samples) {
        e.rel_trans /= e.rel_samples;
        e.rel_rot /= e.rel_samples;
    }
    return e;
}

}  // namespace kitti_io
}  // namespace keyframe_bundle_adjustment
