// landmark_init.hpp — first 3-D position of a landmark from its measurements (one item per landmark).
//
// Replaces BundleAdjusterKeyframes::calculateLandmark (both overloads),
//   keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:332-355  depth back-projection of the first measurement
//                                                                        with d >= 0,
//   :358-382 + internal/triangulator.hpp:51-75                           N-view midpoint triangulation: the point
//                                                                        minimising the distances to all rays,
//                                                                        sum_i (I - r_i r_i^T) p = sum_i (I - r_i r_i^T) c_i,
//                                                                        solved like Eigen's JacobiSVD (minimum norm),
//   convertMeasurementToRay, src/definitions.cpp:98-102.
// Same statements on gfx950 (landmark_init.hip, one lane per landmark) and on the host (the emulated ABI of the
// CPU test tier) - and the same statements, in the same order, as the oracle's restatement of the reference lines
// (oracle/exact_oracle.cpp:oracle_landmark_init).  BIT-EXACT CONTRACT: landmark_init.hip is compiled with floating-point
// contraction off, so the positions equal the oracle's bit for bit - a two-view triangulation at 0.5 m baseline amplifies
// the last bits by its condition number, and the positions feed the landmark selection (cheirality, voxel cells,
// distance ranks), whose decisions would otherwise differ between two drives of the same data.
#pragma once
#include "../../include/limo_hip.h"
#include "kba_math.hpp"

namespace kba {

// camera<-origin pose (7) -> origin<-camera rotation (row-major) and camera centre in the origin frame
KBA_HD void lminit_invert_pose(const double* pose_cam_origin, double* R_oc, double* c_o) {
    double R[9];
    quat_R(pose_cam_origin, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R_oc[i * 3 + j] = R[j * 3 + i];
    const double* t = pose_cam_origin + 4;
    for (int i = 0; i < 3; ++i) c_o[i] = -(R_oc[i * 3] * t[0] + R_oc[i * 3 + 1] * t[1] + R_oc[i * 3 + 2] * t[2]);
}

// Minimum-norm least-squares solution of the symmetric PSD 3x3 system A p = b (what JacobiSVD::solve returns), via
// a cyclic-Jacobi eigen-decomposition; eigenvalues below eps * 3 * max are treated as zero.
KBA_HD void lminit_solve_sym3_pinv(const double* A, const double* b, double* p) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {A[6], A[7], A[8]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                if (a[i][j] == 0.0) continue;
                const double tau = (a[j][j] - a[i][i]) / (2.0 * a[i][j]);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
                for (int k = 0; k < 3; ++k) {  // A <- A G
                    const double x = a[k][i], y = a[k][j];
                    a[k][i] = cs * x - sn * y;
                    a[k][j] = sn * x + cs * y;
                }
                for (int k = 0; k < 3; ++k) {  // A <- G^T A
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
    const double mx = fmax(fabs(a[0][0]), fmax(fabs(a[1][1]), fabs(a[2][2])));
    const double thr = 2.220446049250313e-16 * 3.0 * mx;
    p[0] = p[1] = p[2] = 0.0;
    for (int j = 0; j < 3; ++j) {
        if (fabs(a[j][j]) <= thr) continue;
        const double dot = V[0][j] * b[0] + V[1][j] * b[1] + V[2][j] * b[2];
        for (int k = 0; k < 3; ++k) p[k] += V[k][j] * dot / a[j][j];
    }
}

// Landmark i of the CSR ray list.  Returns false (ok = 0) where no position can be computed.
KBA_HD bool lminit_one(const int32_t* ray_off, const limo_ray* rays, const uint8_t* use_depth, int i, double* pos) {
    const int b = ray_off[i], e = ray_off[i + 1];
    if (use_depth[i]) {
        for (int r = b; r < e; ++r) {  // first measurement with d >= 0 is back-projected (:336-351)
            const limo_ray& m = rays[r];
            if (m.d < 0) continue;
            const double z = static_cast<double>(m.d);
            const double pc[3] = {(static_cast<double>(m.u) - m.cx) * z / m.f, (static_cast<double>(m.v) - m.cy) * z / m.f, z};
            double R_oc[9], c_o[3];
            lminit_invert_pose(m.pose_cam_origin, R_oc, c_o);
            for (int k = 0; k < 3; ++k) pos[k] = R_oc[k * 3] * pc[0] + R_oc[k * 3 + 1] * pc[1] + R_oc[k * 3 + 2] * pc[2] + c_o[k];
            return true;
        }
        return false;
    }
    if (e - b < 2) return false;  // :363-365
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rhs[3] = {0, 0, 0};
    for (int r = b; r < e; ++r) {
        const limo_ray& m = rays[r];
        // convertMeasurementToRay, definitions.cpp:98-102: (K^-1 (u,v,1)).normalized()
        double ray[3] = {(static_cast<double>(m.u) - m.cx) / m.f, (static_cast<double>(m.v) - m.cy) / m.f, 1.0};
        const double nn = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
        for (int k = 0; k < 3; ++k) ray[k] /= nn;
        double R_oc[9], c_o[3], ro[3];
        lminit_invert_pose(m.pose_cam_origin, R_oc, c_o);
        for (int k = 0; k < 3; ++k) ro[k] = R_oc[k * 3] * ray[0] + R_oc[k * 3 + 1] * ray[1] + R_oc[k * 3 + 2] * ray[2];
        double cur[9];  // I - r r^T (triangulator.hpp:60-66)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) cur[a * 3 + c] = (a == c ? 1.0 : 0.0) - ro[a] * ro[c];
        for (int k = 0; k < 9; ++k) A[k] += cur[k];
        for (int a = 0; a < 3; ++a) rhs[a] += cur[a * 3] * c_o[0] + cur[a * 3 + 1] * c_o[1] + cur[a * 3 + 2] * c_o[2];
    }
    lminit_solve_sym3_pinv(A, rhs, pos);
    return true;
}

}  // namespace kba
