// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.hpp header).
//
// exact_oracle.cpp — the parts of the BA oracle the HIP path is held to BIT FOR BIT: landmark initialisation
// (BundleAdjusterKeyframes::calculateLandmark, keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:332-382,
// internal/triangulator.hpp:51-75, convertMeasurementToRay src/definitions.cpp:98-102).  Compiled with
// -ffp-contract=off and without host-specific instruction selection (oracle/Makefile) like depth_oracle.cpp: every
// + - * / sqrt is one IEEE-754 operation in the order written, so limo_amd/csrc/landmark_init.hpp (same statements,
// same order) produces the same bits on gfx950.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

#include "../include/limo_hip.h"
#include "functors.hpp"

using namespace kba_oracle;

namespace {

// symmetric 3x3 eigen decomposition by cyclic Jacobi (for the JacobiSVD::solve of triangulator.hpp:71)
void sym3_eig(const double A[9], double evals[3], double V[9]) {
    double a[9];
    for (int i = 0; i < 9; ++i) a[i] = A[i];
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p * 3 + q] == 0.0) continue;
                double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * a[p * 3 + q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - s * akq;
                    a[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - s * aqk;
                    a[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) evals[i] = a[i * 3 + i];
}

}  // namespace

extern "C" {

// calculateLandmark (both overloads), bundle_adjuster_keyframes.cpp:332-382 + triangulator.hpp:51-75
int oracle_landmark_init(int32_t n, const int32_t* ray_off, const limo_ray* rays, const uint8_t* use_depth,
                         double* pos_out, uint8_t* ok) {
    for (int i = 0; i < n; ++i) {
        const int b = ray_off[i], e = ray_off[i + 1];
        ok[i] = 0;
        if (use_depth[i]) {
            for (int r = b; r < e; ++r) {
                if (rays[r].d < 0) continue;  // :338
                const double z = static_cast<double>(rays[r].d);
                const double x = (static_cast<double>(rays[r].u) - rays[r].cx) * z / rays[r].f;
                const double y = (static_cast<double>(rays[r].v) - rays[r].cy) * z / rays[r].f;
                Iso<double> T = inverse(convert(rays[r].pose_cam_origin));
                double p[3] = {x, y, z};
                apply(T, p, pos_out + 3 * i);
                ok[i] = 1;
                break;
            }
        } else {
            if (e - b < 2) continue;  // :363-365
            double sum_rrt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rhs[3] = {0, 0, 0};
            for (int r = b; r < e; ++r) {
                // convertMeasurementToRay, definitions.cpp:98-102: (K^-1 (u,v,1)).normalized()
                double ray[3] = {(static_cast<double>(rays[r].u) - rays[r].cx) / rays[r].f,
                                 (static_cast<double>(rays[r].v) - rays[r].cy) / rays[r].f, 1.0};
                double nn = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
                for (int q = 0; q < 3; ++q) ray[q] /= nn;
                Iso<double> T = inverse(convert(rays[r].pose_cam_origin));  // pose_origin_camera
                double rt[3];
                for (int q = 0; q < 3; ++q) rt[q] = T.R[3 * q] * ray[0] + T.R[3 * q + 1] * ray[1] + T.R[3 * q + 2] * ray[2];
                double cur[9];
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) cur[a * 3 + c] = (a == c ? 1.0 : 0.0) - rt[a] * rt[c];
                for (int q = 0; q < 9; ++q) sum_rrt[q] += cur[q];
                for (int a = 0; a < 3; ++a) rhs[a] += cur[a * 3] * T.t[0] + cur[a * 3 + 1] * T.t[1] + cur[a * 3 + 2] * T.t[2];
            }
            // jacobiSvd(FullU|FullV).solve(rhs): pseudo-inverse with Eigen's default threshold eps*3*max_sv
            double ev[3], V[9];
            sym3_eig(sum_rrt, ev, V);
            double mx = std::max(std::fabs(ev[0]), std::max(std::fabs(ev[1]), std::fabs(ev[2])));
            double thr = std::numeric_limits<double>::epsilon() * 3.0 * mx;
            double p[3] = {0, 0, 0};
            for (int j = 0; j < 3; ++j) {
                if (std::fabs(ev[j]) <= thr) continue;
                double dot = V[0 * 3 + j] * rhs[0] + V[1 * 3 + j] * rhs[1] + V[2 * 3 + j] * rhs[2];
                for (int a = 0; a < 3; ++a) p[a] += V[a * 3 + j] * dot / ev[j];
            }
            for (int a = 0; a < 3; ++a) pos_out[3 * i + a] = p[a];
            ok[i] = 1;
        }
    }
    return LIMO_OK;
}

}  // extern "C"
