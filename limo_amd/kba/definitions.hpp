// definitions.hpp — data types of the keyframe_bundle_adjustment API surface, Eigen-free.
//
// Mirrors (names, members, meaning) the reference headers
//   matches_msg_types/include/matches_msg_types/{feature_point,tracklet,tracklets}.hpp
//   keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/internal/definitions.hpp:13-169
// so that code written against the reference (its tests, its ROS-free driver) compiles against this shim with the
// HIP library behind it.  Eigen is not available in this environment; `EigenPose`, `Vector2d`, `Vector3d` are small
// value types with the handful of operations the API uses (compose, inverse, apply, matrix access).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <vector>

namespace matches_msg_types {

struct FeaturePoint {  // feature_point.hpp:4-36
    FeaturePoint() : u(0), v(0), d(-1) {}
    FeaturePoint(float u_, float v_) : u(u_), v(v_), d(-1) {}
    FeaturePoint(float u_, float v_, float d_) : u(u_), v(v_), d(d_) {}
    float u;
    float v;
    float d;  // metres along the camera z axis; < 0 = no depth
};

struct Tracklet {  // tracklet.hpp:5-12
    std::vector<FeaturePoint> feature_points;
    unsigned long id = 0;
    unsigned long age = 0;
    bool is_outlier{false};
    int label{-2};
};

using TimestampNSec = uint64_t;

struct Tracklets {  // tracklets.hpp:10-13
    std::vector<TimestampNSec> stamps;
    std::vector<Tracklet> tracks;
};

}  // namespace matches_msg_types

namespace keyframe_bundle_adjustment {

struct Vector2d {
    double v[2]{0, 0};
    Vector2d() = default;
    Vector2d(double x, double y) : v{x, y} {}
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
};

struct Vector3d {
    double v[3]{0, 0, 0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    explicit Vector3d(const double* p) : v{p[0], p[1], p[2]} {}
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
};

// Rigid transform x' = R x + t (what the reference spells Eigen::Isometry3d).
struct EigenPose {
    double R[9]{1, 0, 0, 0, 1, 0, 0, 0, 1};
    double t[3]{0, 0, 0};
    static EigenPose Identity() { return EigenPose(); }
    Vector3d translation() const { return {t[0], t[1], t[2]}; }
    EigenPose operator*(const EigenPose& b) const {
        EigenPose c;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) c.R[3 * i + j] = R[3 * i] * b.R[j] + R[3 * i + 1] * b.R[3 + j] + R[3 * i + 2] * b.R[6 + j];
            c.t[i] = R[3 * i] * b.t[0] + R[3 * i + 1] * b.t[1] + R[3 * i + 2] * b.t[2] + t[i];
        }
        return c;
    }
    Vector3d operator*(const Vector3d& p) const {
        return {R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0], R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1],
                R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2]};
    }
    EigenPose inverse() const {
        EigenPose c;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) c.R[3 * i + j] = R[3 * j + i];
        for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * t[0] + c.R[3 * i + 1] * t[1] + c.R[3 * i + 2] * t[2]);
        return c;
    }
    // Eigen-style in-place builders used by the reference tests: p.translate(v); p.rotate(axis-angle)
    EigenPose& translate(const Vector3d& v) {  // this = this * Translation(v)
        for (int i = 0; i < 3; ++i) t[i] += R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
        return *this;
    }
    EigenPose& rotate(double angle, const Vector3d& axis) {  // this = this * AngleAxis(angle, axis)
        const double n = axis.norm();
        const double x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = std::cos(angle), s = std::sin(angle), C = 1 - c;
        const double Q[9] = {c + x * x * C, x * y * C - z * s, x * z * C + y * s, y * x * C + z * s, c + y * y * C,
                             y * z * C - x * s, z * x * C - y * s, z * y * C + x * s, c + z * z * C};
        double N[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) N[3 * i + j] = R[3 * i] * Q[j] + R[3 * i + 1] * Q[3 + j] + R[3 * i + 2] * Q[6 + j];
        for (int i = 0; i < 9; ++i) R[i] = N[i];
        return *this;
    }
    // Eigen's isApprox on the 4x4 matrix: |a - b|_F <= prec * min(|a|_F, |b|_F)
    bool isApprox(const EigenPose& o, double prec = 1e-12) const {
        double d2 = 0, a2 = 1.0, b2 = 1.0;  // the (3,3) entry is 1 in both
        for (int i = 0; i < 9; ++i) {
            d2 += (R[i] - o.R[i]) * (R[i] - o.R[i]);
            a2 += R[i] * R[i];
            b2 += o.R[i] * o.R[i];
        }
        for (int i = 0; i < 3; ++i) {
            d2 += (t[i] - o.t[i]) * (t[i] - o.t[i]);
            a2 += t[i] * t[i];
            b2 += o.t[i] * o.t[i];
        }
        return d2 <= prec * prec * std::min(a2, b2);
    }
};

using CameraId = unsigned long;
using TimestampNSec = matches_msg_types::TimestampNSec;
using TimestampSec = double;
using LandmarkId = unsigned long;
using KeyframeId = unsigned long;
using CameraIds = std::vector<CameraId>;
using PoseId = KeyframeId;
using Pose = std::array<double, 7>;  // quaternion (w,x,y,z) and translation (x,y,z)
using Direction = std::array<double, 3>;

struct Plane {  // definitions.hpp:27-34
    Plane() {
        direction = std::array<double, 3>{{0., 0., 1.}};
        distance = -std::numeric_limits<double>::max();  // negative distance means no gp will be used in optimization
    }
    Direction direction;
    double distance;
};

using FeaturePoint = matches_msg_types::FeaturePoint;
using Tracklet = matches_msg_types::Tracklet;
using Tracklets = matches_msg_types::Tracklets;
using Measurement = FeaturePoint;

struct Landmark {  // definitions.hpp:42-68
    using Ptr = std::shared_ptr<Landmark>;
    using ConstPtr = std::shared_ptr<const Landmark>;
    Landmark() {}
    Landmark(const Vector3d& p, bool has_depth = false) : has_measured_depth(has_depth) {
        pos[0] = p[0];
        pos[1] = p[1];
        pos[2] = p[2];
    }
    std::array<double, 3> pos{{0, 0, 0}};
    bool has_measured_depth{false};
    bool is_ground_plane{false};
    double weight{1.};
};

Pose convert(const EigenPose& p);           // definitions.cpp:14-28 (rotation matrix -> unit quaternion)
EigenPose convert(const Pose& pose);        // definitions.hpp:75-88
TimestampSec convert(const TimestampNSec& ts);
TimestampNSec convert(const TimestampSec& ts);
double calcQuaternionDiff(const Pose& p0, const Pose& p1);  // definitions.cpp:104-111

struct Camera {  // definitions.hpp:93-124
    using Ptr = std::shared_ptr<Camera>;
    Camera(double f, const Vector2d& pp, const EigenPose& pose_cam_veh) : focal_length(f), principal_point(pp) {
        pose_camera_vehicle = convert(pose_cam_veh);
    }
    EigenPose getEigenPose() const { return convert(pose_camera_vehicle); }
    Vector3d getViewingRay(const Measurement& m) const {
        Vector3d r((static_cast<double>(m.u) - principal_point[0]) / focal_length,
                   (static_cast<double>(m.v) - principal_point[1]) / focal_length, 1.0);
        return r * (1.0 / r.norm());
    }
    double focal_length;
    Vector2d principal_point;
    Pose pose_camera_vehicle;  // camera <- vehicle
};

}  // namespace keyframe_bundle_adjustment
