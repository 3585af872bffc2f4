// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// functors.hpp — line-by-line restatement of the reference's residual functors, templated on the scalar so
// the same body runs with double (values) and Jet<N> (derivatives), like ceres::AutoDiffCostFunction does.
// Source followed: keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/internal/cost_functors_ceres.hpp
// and internal/definitions.hpp:75-88 (pose algebra).  Eigen (3.3.4, not in the tree) is restated only for the
// operations those lines use: Quaternion::toRotationMatrix (un-normalised polynomial form),
// Transform<Isometry>::operator*, ::inverse(), ::translation(), Matrix::norm()/normalize().
#pragma once
#include <array>

#include "jet.hpp"

namespace kba_oracle {

// Eigen::Transform<T,3,Isometry> restated: linear part R (row-major 3x3) and translation t.
template <typename T>
struct Iso {
    T R[9];
    T t[3];
};

// definitions.hpp:75-83  convert(const T* pose): p = Identity; p.translate(t); p.rotate(Quaternion(w,x,y,z))
// Eigen's QuaternionBase::toRotationMatrix does NOT normalise q.
template <typename T>
inline Iso<T> convert(const T* pose) {
    const T w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const T tx = T(2.0) * x, ty = T(2.0) * y, tz = T(2.0) * z;
    const T twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x;
    const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Iso<T> p;
    p.R[0] = T(1.0) - (tyy + tzz);
    p.R[1] = txy - twz;
    p.R[2] = txz + twy;
    p.R[3] = txy + twz;
    p.R[4] = T(1.0) - (txx + tzz);
    p.R[5] = tyz - twx;
    p.R[6] = txz - twy;
    p.R[7] = tyz + twx;
    p.R[8] = T(1.0) - (txx + tyy);
    p.t[0] = pose[4];
    p.t[1] = pose[5];
    p.t[2] = pose[6];
    return p;
}

template <typename T>
inline Iso<T> compose(const Iso<T>& a, const Iso<T>& b) {  // a * b
    Iso<T> c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            c.R[3 * i + j] = a.R[3 * i + 0] * b.R[0 + j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        }
        c.t[i] = a.R[3 * i + 0] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return c;
}

template <typename T>
inline Iso<T> inverse(const Iso<T>& a) {  // Isometry: (R^T, -R^T t)
    Iso<T> c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i + 0] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
    return c;
}

template <typename T>
inline void apply(const Iso<T>& a, const T* p, T* out) {
    for (int i = 0; i < 3; ++i) out[i] = a.R[3 * i + 0] * p[0] + a.R[3 * i + 1] * p[1] + a.R[3 * i + 2] * p[2] + a.t[i];
}

template <typename T>
inline Iso<T> cast_iso(const Iso<double>& a) {
    Iso<T> c;
    for (int i = 0; i < 9; ++i) c.R[i] = T(a.R[i]);
    for (int i = 0; i < 3; ++i) c.t[i] = T(a.t[i]);
    return c;
}

using Pose7 = std::array<double, 7>;

// cost_functors_ceres.hpp:53-182 (compensate_rotation is always false on this path, bundle_adjuster_keyframes.cpp:825)
struct ReprojectionErrorWithQuaternions {
    double observed_x, observed_y, focal_length, principal_point_x, principal_point_y;
    Pose7 pose_C_X;
    static constexpr int kNumResiduals = 2;

    template <typename T>
    bool project(const T* point_C, T& predicted_x, T& predicted_y) const {  // :71-89
        T x_cam = point_C[0];
        T y_cam = point_C[1];
        if (jabs(point_C[2]) >= T(0.01)) {
            x_cam /= point_C[2];
            y_cam /= point_C[2];
        } else {
            return false;
        }
        predicted_x = T(focal_length) * x_cam + T(principal_point_x);
        predicted_y = T(focal_length) * y_cam + T(principal_point_y);
        return true;
    }

    template <typename T>
    bool operator()(const T* const pose_X_O, const T* const point_O, T* residuals) const {  // :91-155
        T pose_C_X_cast[7];
        for (int i = 0; i < 7; ++i) pose_C_X_cast[i] = T(pose_C_X[i]);
        Iso<T> pose_C_X_eigen = convert(pose_C_X_cast);
        Iso<T> pose_X_O_eigen = convert(pose_X_O);
        T point_C[3];
        apply(compose(pose_C_X_eigen, pose_X_O_eigen), point_O, point_C);  // :122
        T predicted_x, predicted_y;
        if (!project(point_C, predicted_x, predicted_y)) return false;
        residuals[0] = (predicted_x - T(observed_x));  // rot_comp == 1
        residuals[1] = (predicted_y - T(observed_y));
        return true;
    }
};

// cost_functors_ceres.hpp:187-222
struct LandmarkDepthError {
    double depth_;
    Pose7 pose_C_X_;
    static constexpr int kNumResiduals = 1;
    template <typename T>
    bool operator()(const T* const pose_X_O, const T* const point_O, T* residuals) const {
        T pose_C_X_cast[7];
        for (int i = 0; i < 7; ++i) pose_C_X_cast[i] = T(pose_C_X_[i]);
        Iso<T> pose_C_X_eigen = convert(pose_C_X_cast);
        Iso<T> pose_X_O_eigen = convert(pose_X_O);
        T point_C[3];
        apply(compose(pose_C_X_eigen, pose_X_O_eigen), point_O, point_C);
        residuals[0] = point_C[2] - T(depth_);
        return true;
    }
};

// cost_functors_ceres.hpp:224-250
struct PoseRegularization {
    double scale_;
    static constexpr int kNumResiduals = 1;
    template <typename T>
    bool operator()(const T* const pose1, const T* const pose0, T* residuals) const {
        Iso<T> pose1_eigen = convert(pose1);
        Iso<T> pose0_eigen = convert(pose0);
        Iso<T> diff = compose(pose1_eigen, inverse(pose0_eigen));
        T res = jsqrt(diff.t[0] * diff.t[0] + diff.t[1] * diff.t[1] + diff.t[2] * diff.t[2]);
        residuals[0] = res - T(scale_);
        return true;
    }
};

// cost_functors_ceres.hpp:300-353
struct SpeedRegularizationVector2 {
    double vel_before_before2_[3];
    double dt_cur_;
    Iso<double> pose_origin_before_eigen_;
    static constexpr int kNumResiduals = 3;
    // ctor :302-317
    static bool make(double ts_cur, double ts_before, double ts_before2, const Pose7& pose_before,
                     const Pose7& pose_before2, SpeedRegularizationVector2& out) {
        out.dt_cur_ = ts_cur - ts_before;
        double dt_before = ts_before - ts_before2;
        if (out.dt_cur_ <= 0. || dt_before <= 0.) return false;  // reference throws std::runtime_error
        Iso<double> pb = convert(pose_before.data());
        Iso<double> pb2 = convert(pose_before2.data());
        Iso<double> pbb2 = compose(pb, inverse(pb2));
        for (int i = 0; i < 3; ++i) out.vel_before_before2_[i] = pbb2.t[i] / dt_before;
        out.pose_origin_before_eigen_ = inverse(pb);
        return true;
    }
    template <typename T>
    bool operator()(const T* const pose_cur_origin, T* residuals) const {
        Iso<T> pose_cur_origin_eigen = convert(pose_cur_origin);
        Iso<T> pose_cur_before = compose(pose_cur_origin_eigen, cast_iso<T>(pose_origin_before_eigen_));
        for (int i = 0; i < 3; ++i) residuals[i] = pose_cur_before.t[i] / T(dt_cur_) - T(vel_before_before2_[i]);
        return true;
    }
};

// cost_functors_ceres.hpp:355-392
struct GroundPlaneHeightRegularization {
    static constexpr int kNumResiduals = 1;
    template <typename T>
    bool operator()(const T* const pose_X_O, const T* const plane_dir, const T* const dist, const T* const point_O,
                    T* residuals) const {
        Iso<T> pose_X_O_eigen = convert(pose_X_O);
        T point_X[3];
        apply(pose_X_O_eigen, point_O, point_X);
        residuals[0] = plane_dir[0] * point_X[0] + plane_dir[1] * point_X[1] + plane_dir[2] * point_X[2] + dist[0];
        return true;
    }
};

// cost_functors_ceres.hpp:394-414
struct VectorDifferenceRegularization {
    static constexpr int kNumResiduals = 3;
    template <typename T>
    bool operator()(const T* const plane_dir0, const T* const plane_dir1, T* residuals) const {
        residuals[0] = plane_dir0[0] - plane_dir1[0];
        residuals[1] = plane_dir0[1] - plane_dir1[1];
        residuals[2] = plane_dir0[2] - plane_dir1[2];
        return true;
    }
};

// cost_functors_ceres.hpp:416-438
struct VectorDifferenceRegularization2 {
    double plane_dir0_[3];
    static constexpr int kNumResiduals = 3;
    template <typename T>
    bool operator()(const T* const plane_dir1, T* residuals) const {
        residuals[0] = T(plane_dir0_[0]) - plane_dir1[0];
        residuals[1] = T(plane_dir0_[1]) - plane_dir1[1];
        residuals[2] = T(plane_dir0_[2]) - plane_dir1[2];
        return true;
    }
};

// cost_functors_ceres.hpp:440-469 (dead in the pipeline; pinned by a reference KAT)
struct TranslationDifferenceRegularization {
    static constexpr int kNumResiduals = 3;
    template <typename T>
    bool operator()(const T* const pose0, const T* const pose1, const T* const pose2, T* residuals) const {
        Iso<T> p0 = convert(pose0), p1 = convert(pose1), p2 = convert(pose2);
        Iso<T> diff10 = compose(p1, inverse(p0));
        Iso<T> diff21 = compose(p2, inverse(p1));
        for (int i = 0; i < 3; ++i) residuals[i] = diff21.t[i] - diff10.t[i];
        return true;
    }
};

// cost_functors_ceres.hpp:507-526
struct GroundPlaneDistanceRegularization {
    static constexpr int kNumResiduals = 1;
    template <typename T>
    bool operator()(const T* const dist0, const T* const dist1, T* residuals) const {
        residuals[0] = dist0[0] - dist1[0];
        return true;
    }
};

// cost_functors_ceres.hpp:528-555
struct GroundPlaneMotionRegularization {
    static constexpr int kNumResiduals = 1;
    template <typename T>
    bool operator()(const T* const pose_0, const T* const pose_1, const T* const plane_dir0, T* residuals) const {
        Iso<T> pose_eigen_0 = convert(pose_0);
        Iso<T> pose_eigen_1 = convert(pose_1);
        Iso<T> d = compose(pose_eigen_0, inverse(pose_eigen_1));
        T delta_trans[3] = {d.t[0], d.t[1], d.t[2]};
        // Eigen normalize(): divide by sqrt(squaredNorm) when squaredNorm > 0
        T z = delta_trans[0] * delta_trans[0] + delta_trans[1] * delta_trans[1] + delta_trans[2] * delta_trans[2];
        if (scalar_of(z) > 0.0) {
            T n = jsqrt(z);
            for (int i = 0; i < 3; ++i) delta_trans[i] = delta_trans[i] / n;
        }
        residuals[0] = plane_dir0[0] * delta_trans[0] + plane_dir0[1] * delta_trans[1] + plane_dir0[2] * delta_trans[2];
        return true;
    }
};

// local_parameterizations.hpp:135-165
struct FixScaleVectorPlus {
    double scale_ = 1.0;
    template <typename T>
    bool operator()(const T* x, const T* delta, T* x_plus_delta) const {
        x_plus_delta[0] = x[0] + delta[0];
        x_plus_delta[1] = x[1] + delta[1];
        x_plus_delta[2] = x[2] + delta[2];
        T norm = jsqrt(x_plus_delta[0] * x_plus_delta[0] + x_plus_delta[1] * x_plus_delta[1] +
                       x_plus_delta[2] * x_plus_delta[2]);
        T factor = T(scale_) / norm;
        x_plus_delta[0] *= factor;
        x_plus_delta[1] *= factor;
        x_plus_delta[2] *= factor;
        return true;
    }
};

}  // namespace kba_oracle
