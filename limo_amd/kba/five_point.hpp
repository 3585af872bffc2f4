// five_point.hpp — relative camera motion from image correspondences alone: the motion prior the node falls back to when no
// external prior exists (SURVEY §8f-4 "optional").  Restates what the reference gets from OpenCV at
// keyframe_bundle_adjustment_ros_tool/src/commons/general_helpers.hpp:103-140 (calcMotion5Point: mean-flow gate,
// cv::findEssentialMat(points1, points0, focal, pp, RANSAC, probability, 2.0), cv::recoverPose) and :209-231
// (getMotionUnscaled: forward unit translation when the gate fails, scaling by speed x dt, camera -> vehicle frame), without
// OpenCV / Eigen:
//   * minimal solver: the five-point problem in its standard form - E in the 4-dimensional null space of the five epipolar
//     constraints, E = x X + y Y + z Z + W; det E = 0 and 2 E E^T E - tr(E E^T) E = 0 give ten cubics in (x, y, z); Gauss-Jordan
//     on their 10 x 20 coefficient matrix leaves a 10 x 10 action matrix of "multiply by x" on the monomials
//     [x^2 xy xz y^2 yz z^2 x y z 1], whose real eigenpairs are the solutions (Nister 2004 / Stewenius et al. 2006).  Eigenvalues by
//     Hessenberg reduction + double-shift QR, the real ones polished together with their eigenvector by inverse iteration (a
//     characteristic polynomial loses them: the matrix has entries of 1e3-1e4, its coefficients span 40 orders of magnitude);
//   * RANSAC over minimal samples, Sampson distance against `threshold_px / focal`, adaptive number of samples for `probability`,
//     seeded (OpenCV draws from its global RNG: the choice of samples differs, the model family and the inlier rule do not);
//   * pose from E: the four (R, t) decompositions, the one that puts most inliers in front of both cameras.
// Host code, header only.  Conventions follow OpenCV's: for points a (first argument) and b (second), b^T E a = 0 and
// x_b = R x_a + t in camera coordinates, |t| = 1.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <set>
#include <vector>

#include "definitions.hpp"

namespace keyframe_bundle_adjustment {
namespace five_point {

using Mat3 = std::array<double, 9>;  // row-major

namespace detail {

// ---- polynomials in (x, y, z) of total degree <= 1 / 2 / 3 as coefficient vectors over fixed monomial lists
//      degree 1: [x y z 1]; degree 2: [x2 xy xz y2 yz z2 x y z 1]; degree 3: [x3 x2y x2z xy2 xyz xz2 y3 y2z yz2 z3 | degree-2 list]
struct Exp {
    int i, j, k;
};
constexpr Exp kM1[4] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
constexpr Exp kM2[10] = {{2, 0, 0}, {1, 1, 0}, {1, 0, 1}, {0, 2, 0}, {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
constexpr Exp kM3[20] = {{3, 0, 0}, {2, 1, 0}, {2, 0, 1}, {1, 2, 0}, {1, 1, 1}, {1, 0, 2}, {0, 3, 0}, {0, 2, 1}, {0, 1, 2}, {0, 0, 3},
                         {2, 0, 0}, {1, 1, 0}, {1, 0, 1}, {0, 2, 0}, {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
inline int index2(const Exp& e) {
    for (int q = 0; q < 10; ++q)
        if (kM2[q].i == e.i && kM2[q].j == e.j && kM2[q].k == e.k) return q;
    return -1;
}
inline int index3(const Exp& e) {
    for (int q = 0; q < 20; ++q)
        if (kM3[q].i == e.i && kM3[q].j == e.j && kM3[q].k == e.k) return q;
    return -1;
}
using P1 = std::array<double, 4>;
using P2 = std::array<double, 10>;
using P3 = std::array<double, 20>;
inline P2 mul11(const P1& a, const P1& b) {
    static const auto table = [] {
        std::array<std::array<int, 4>, 4> t{};
        for (int p = 0; p < 4; ++p)
            for (int q = 0; q < 4; ++q) t[p][q] = index2({kM1[p].i + kM1[q].i, kM1[p].j + kM1[q].j, kM1[p].k + kM1[q].k});
        return t;
    }();
    P2 c{};
    for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 4; ++q) c[table[p][q]] += a[p] * b[q];
    return c;
}
inline P3 mul21(const P2& a, const P1& b) {
    static const auto table = [] {
        std::array<std::array<int, 4>, 10> t{};
        for (int p = 0; p < 10; ++p)
            for (int q = 0; q < 4; ++q) t[p][q] = index3({kM2[p].i + kM1[q].i, kM2[p].j + kM1[q].j, kM2[p].k + kM1[q].k});
        return t;
    }();
    P3 c{};
    for (int p = 0; p < 10; ++p)
        for (int q = 0; q < 4; ++q) c[table[p][q]] += a[p] * b[q];
    return c;
}
template <class P>
P add(const P& a, const P& b, double sb = 1.) {
    P c = a;
    for (size_t q = 0; q < c.size(); ++q) c[q] += sb * b[q];
    return c;
}

// eigenvectors of a symmetric n x n matrix by cyclic Jacobi rotations: columns of V, eigenvalues ascending
template <int N>
void symmetric_eigen(std::array<double, N * N> a, std::array<double, N>& w, std::array<double, N * N>& V) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1. : 0.;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.;
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) off += a[i * N + j] * a[i * N + j];
        double diag = 0.;
        for (int i = 0; i < N; ++i) diag += a[i * N + i] * a[i * N + i];
        if (off <= 1e-32 * diag || off < 1e-300) break;  // (off-diagonal mass below the rounding level of the diagonal)
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                if (std::fabs(a[p * N + q]) < 1e-300) continue;
                const double theta = (a[q * N + q] - a[p * N + p]) / (2. * a[p * N + q]);
                const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < N; ++k) {  // A <- A J
                    const double akp = a[k * N + p], akq = a[k * N + q];
                    a[k * N + p] = c * akp - s * akq;
                    a[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {  // A <- J^T A
                    const double apk = a[p * N + k], aqk = a[q * N + k];
                    a[p * N + k] = c * apk - s * aqk;
                    a[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
    std::array<int, N> order;
    for (int i = 0; i < N; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a[x * N + x] < a[y * N + y]; });
    std::array<double, N * N> Vs;
    for (int c = 0; c < N; ++c) {
        w[c] = a[order[c] * N + order[c]];
        for (int r = 0; r < N; ++r) Vs[r * N + c] = V[r * N + order[c]];
    }
    V = Vs;
}

// solve (A - lambda I) v = rhs in place by Gaussian elimination with partial pivoting (10 x 10); false if singular to working precision
inline bool solve10(std::array<double, 100> a, std::array<double, 10>& b) {
    for (int c = 0; c < 10; ++c) {
        int piv = c;
        for (int r = c + 1; r < 10; ++r)
            if (std::fabs(a[r * 10 + c]) > std::fabs(a[piv * 10 + c])) piv = r;
        if (std::fabs(a[piv * 10 + c]) < 1e-300) return false;
        if (piv != c) {
            for (int k = 0; k < 10; ++k) std::swap(a[piv * 10 + k], a[c * 10 + k]);
            std::swap(b[piv], b[c]);
        }
        for (int r = c + 1; r < 10; ++r) {
            const double f = a[r * 10 + c] / a[c * 10 + c];
            if (f == 0.) continue;
            for (int k = c; k < 10; ++k) a[r * 10 + k] -= f * a[c * 10 + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 9; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < 10; ++k) s -= a[r * 10 + k] * b[k];
        b[r] = s / a[r * 10 + r];
    }
    return true;
}

// Eigenvalues of a real 10 x 10 matrix: reduction to upper Hessenberg form by stabilised elementary transformations, then the
// double-shift QR iteration (Francis) on the active block with deflation - the algorithm EISPACK calls elmhes + hqr.  Returns
// false if an eigenvalue does not settle within 60 iterations.
inline bool eigenvalues10(std::array<double, 100> a, std::array<double, 10>& wr, std::array<double, 10>& wi) {
    constexpr int n = 10;
    auto A = [&a](int i, int j) -> double& { return a[i * n + j]; };
    // ---- Hessenberg form
    for (int m = 1; m < n - 1; ++m) {
        double x = 0.;
        int piv = m;
        for (int j = m; j < n; ++j)
            if (std::fabs(A(j, m - 1)) > std::fabs(x)) {
                x = A(j, m - 1);
                piv = j;
            }
        if (piv != m) {
            for (int j = m - 1; j < n; ++j) std::swap(A(piv, j), A(m, j));
            for (int j = 0; j < n; ++j) std::swap(A(j, piv), A(j, m));
        }
        if (x != 0.)
            for (int i = m + 1; i < n; ++i) {
                double y = A(i, m - 1);
                if (y == 0.) continue;
                y /= x;
                A(i, m - 1) = y;
                for (int j = m; j < n; ++j) A(i, j) -= y * A(m, j);
                for (int j = 0; j < n; ++j) A(j, m) += y * A(j, i);
            }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) A(i, j) = 0.;
    // ---- QR iteration
    double anorm = 0.;
    for (int i = 0; i < n; ++i)
        for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::fabs(A(i, j));
    int nn = n - 1;
    double t = 0.;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {  // a negligible subdiagonal element splits the block
                double s = std::fabs(A(l - 1, l - 1)) + std::fabs(A(l, l));
                if (s == 0.) s = anorm;
                if (std::fabs(A(l, l - 1)) + s == s) {
                    A(l, l - 1) = 0.;
                    break;
                }
            }
            double x = A(nn, nn);
            if (l == nn) {  // one real eigenvalue
                wr[nn] = x + t;
                wi[nn--] = 0.;
            } else {
                double y = A(nn - 1, nn - 1), w = A(nn, nn - 1) * A(nn - 1, nn);
                if (l == nn - 1) {  // a 2 x 2 block: two eigenvalues
                    const double p = 0.5 * (y - x), q = p * p + w;
                    double z = std::sqrt(std::fabs(q));
                    x += t;
                    if (q >= 0.) {
                        z = p + (p >= 0. ? std::fabs(z) : -std::fabs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn - 1] = -(wi[nn] = z);
                    }
                    nn -= 2;
                } else {  // no deflation yet: one double-shift step
                    if (its == 60) return false;
                    if (its == 10 || its == 20 || its == 30 || its == 40 || its == 50) {  // exceptional shift
                        t += x;
                        for (int i = 0; i <= nn; ++i) A(i, i) -= x;
                        const double s = std::fabs(A(nn, nn - 1)) + std::fabs(A(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    double p = 0., q = 0., r = 0., z;
                    for (m = nn - 2; m >= l; --m) {  // start of the step: two consecutive small subdiagonal elements
                        z = A(m, m);
                        r = x - z;
                        double s = y - z;
                        p = (r * s - w) / A(m + 1, m) + A(m, m + 1);
                        q = A(m + 1, m + 1) - z - r - s;
                        r = A(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s;
                        q /= s;
                        r /= s;
                        if (m == l) break;
                        const double u = std::fabs(A(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        const double v = std::fabs(p) * (std::fabs(A(m - 1, m - 1)) + std::fabs(z) + std::fabs(A(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        A(i, i - 2) = 0.;
                        if (i != m + 2) A(i, i - 3) = 0.;
                    }
                    for (int k = m; k <= nn - 1; ++k) {  // the bulge is chased down the block
                        if (k != m) {
                            p = A(k, k - 1);
                            q = A(k + 1, k - 1);
                            r = k != nn - 1 ? A(k + 2, k - 1) : 0.;
                            x = std::fabs(p) + std::fabs(q) + std::fabs(r);
                            if (x != 0.) {
                                p /= x;
                                q /= x;
                                r /= x;
                            }
                        }
                        const double nrm = std::sqrt(p * p + q * q + r * r);
                        const double s = p >= 0. ? nrm : -nrm;
                        if (s != 0.) {
                            if (k == m) {
                                if (l != m) A(k, k - 1) = -A(k, k - 1);
                            } else {
                                A(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s;
                            y = q / s;
                            z = r / s;
                            q /= p;
                            r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = A(k, j) + q * A(k + 1, j);
                                if (k != nn - 1) {
                                    p += r * A(k + 2, j);
                                    A(k + 2, j) -= p * z;
                                }
                                A(k + 1, j) -= p * y;
                                A(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * A(i, k) + y * A(i, k + 1);
                                if (k != nn - 1) {
                                    p += z * A(i, k + 2);
                                    A(i, k + 2) -= p * r;
                                }
                                A(i, k + 1) -= p * q;
                                A(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return true;
}

inline double det3(const Mat3& m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
inline Mat3 mul3(const Mat3& a, const Mat3& b) {
    Mat3 c{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    return c;
}
inline Mat3 transpose3(const Mat3& a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }

}  // namespace detail

// Essential matrices consistent with five correspondences in NORMALISED image coordinates (b^T E a = 0): up to ten, |E|_F = 1.
// null-space basis (E[e] = coefficients of x, y, z, 1 of entry e) and action matrix; false for a degenerate sample
inline bool actionMatrix(const double a[5][2], const double b[5][2], detail::P1 E[9], std::array<double, 100>& A) {
    using namespace detail;
    // null space of the 5 x 9 constraint matrix: the four eigenvectors of Q^T Q with the smallest eigenvalues
    std::array<double, 81> QtQ{};
    for (int p = 0; p < 5; ++p) {
        const double q[9] = {b[p][0] * a[p][0], b[p][0] * a[p][1], b[p][0], b[p][1] * a[p][0], b[p][1] * a[p][1], b[p][1], a[p][0], a[p][1], 1.};
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) QtQ[i * 9 + j] += q[i] * q[j];
    }
    std::array<double, 9> w;
    std::array<double, 81> V;
    symmetric_eigen<9>(QtQ, w, V);
    // E(x, y, z) = x X + y Y + z Z + W: every entry a degree-1 polynomial
    for (int e = 0; e < 9; ++e) E[e] = {V[e * 9 + 0], V[e * 9 + 1], V[e * 9 + 2], V[e * 9 + 3]};
    // the ten cubics
    std::array<P3, 10> eq;
    {
        const P2 m0 = add(mul11(E[4], E[8]), mul11(E[5], E[7]), -1.), m1 = add(mul11(E[3], E[8]), mul11(E[5], E[6]), -1.),
                 m2 = add(mul11(E[3], E[7]), mul11(E[4], E[6]), -1.);
        eq[0] = add(add(mul21(m0, E[0]), mul21(m1, E[1]), -1.), mul21(m2, E[2]));  // det E
        P2 EEt[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) EEt[3 * i + j] = add(add(mul11(E[3 * i], E[3 * j]), mul11(E[3 * i + 1], E[3 * j + 1])), mul11(E[3 * i + 2], E[3 * j + 2]));
        const P2 tr = add(add(EEt[0], EEt[4]), EEt[8]);
        P2 L[9];  // E E^T - 1/2 tr(E E^T) I
        for (int i = 0; i < 9; ++i) L[i] = EEt[i];
        for (int d : {0, 4, 8}) L[d] = add(L[d], tr, -0.5);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) eq[1 + 3 * i + j] = add(add(mul21(L[3 * i], E[j]), mul21(L[3 * i + 1], E[3 + j])), mul21(L[3 * i + 2], E[6 + j]));
    }
    // Gauss-Jordan on the ten degree-3 columns: [I | B]
    double M[10][20];
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 20; ++c) M[r][c] = eq[r][c];
    for (int c = 0; c < 10; ++c) {
        int piv = c;
        for (int r = c + 1; r < 10; ++r)
            if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
        if (std::fabs(M[piv][c]) < 1e-14) return false;  // degenerate sample
        for (int k = 0; k < 20; ++k) std::swap(M[piv][k], M[c][k]);
        const double inv = 1. / M[c][c];
        for (int k = 0; k < 20; ++k) M[c][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f == 0.) continue;
            for (int k = 0; k < 20; ++k) M[r][k] -= f * M[c][k];
        }
    }
    // action matrix of "multiply by x" on [x2 xy xz y2 yz z2 x y z 1]: x * (x2 .. z2) are the monomials x3 x2y x2z xy2 xyz xz2 = rows 0..5
    A.fill(0.);
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 10; ++c) A[r * 10 + c] = -M[r][10 + c];
    A[6 * 10 + 0] = 1.;  // x * x = x2
    A[7 * 10 + 1] = 1.;  // x * y = xy
    A[8 * 10 + 2] = 1.;  // x * z = xz
    A[9 * 10 + 6] = 1.;  // x * 1 = x
    return true;
}

inline std::vector<Mat3> essentialFromFive(const double a[5][2], const double b[5][2]) {
    using namespace detail;
    P1 E[9];
    std::array<double, 100> A;
    if (!actionMatrix(a, b, E, A)) return {};
    // eigenvalues of the action matrix; the real ones (to a loose tolerance) are polished together with their eigenvector below
    std::array<double, 10> wr, wi;
    if (!eigenvalues10(A, wr, wi)) return {};
    std::vector<std::complex<double>> z(10);
    for (int q = 0; q < 10; ++q) z[q] = {wr[q], wi[q]};
    std::vector<Mat3> out;
    std::vector<double> found;  // eigenvalues already taken
    double a_scale = 0.;
    for (double e : A) a_scale = std::max(a_scale, std::fabs(e));
    for (const auto& root : z) {
        // anything near the real axis is polished on the action matrix by inverse iteration (that also yields the eigenvector),
        // and kept if it settles on a real eigenpair not seen before
        if (std::fabs(root.imag()) > 1e-4 * (1. + std::fabs(root.real()))) continue;
        double lambda = root.real();
        std::array<double, 10> v;
        for (int i = 0; i < 10; ++i) v[i] = 1. / (1. + i);
        bool ok = true;
        for (int it = 0; it < 8 && ok; ++it) {
            std::array<double, 100> S = A;
            const double shift = lambda + 1e-9 * (1. + std::fabs(lambda));  // (an exactly singular system is avoided)
            for (int i = 0; i < 10; ++i) S[i * 10 + i] -= shift;
            std::array<double, 10> u = v;
            ok = solve10(S, u);
            if (!ok) break;
            double uu = 0., uv = 0.;
            for (int i = 0; i < 10; ++i) {
                uu += u[i] * u[i];
                uv += u[i] * v[i];
            }
            if (!(uu > 0.) || !std::isfinite(uu) || uv == 0.) {
                ok = false;
                break;
            }
            // u ~ v / (lambda - shift): eigenvalue estimate from the growth along v
            double vv = 0.;
            for (double x : v) vv += x * x;
            const double lambda_new = shift + vv / uv;
            const bool settled = it >= 1 && std::fabs(lambda_new - lambda) <= 1e-13 * (1. + std::fabs(lambda_new));
            lambda = lambda_new;
            const double n = std::sqrt(uu);
            for (int i = 0; i < 10; ++i) v[i] = u[i] / n;
            if (settled) break;
        }
        if (!ok || !std::isfinite(lambda)) continue;
        double res = 0.;  // |A v - lambda v| with |v| = 1
        for (int i = 0; i < 10; ++i) {
            double sacc = -lambda * v[i];
            for (int q = 0; q < 10; ++q) sacc += A[i * 10 + q] * v[q];
            res += sacc * sacc;
        }
        if (std::sqrt(res) > 1e-7 * (a_scale + std::fabs(lambda))) continue;  // did not settle on a real eigenpair
        bool seen = false;
        for (double f : found) seen = seen || std::fabs(f - lambda) <= 1e-7 * (1. + std::fabs(lambda));
        if (seen || std::fabs(v[9]) < 1e-12) continue;
        found.push_back(lambda);
        const double x = v[6] / v[9], y = v[7] / v[9], zc = v[8] / v[9];
        Mat3 Em;
        double n = 0.;
        for (int e = 0; e < 9; ++e) {
            Em[e] = x * E[e][0] + y * E[e][1] + zc * E[e][2] + E[e][3];
            n += Em[e] * Em[e];
        }
        n = std::sqrt(n);
        if (!(n > 0.) || !std::isfinite(n)) continue;
        for (double& e : Em) e /= n;
        out.push_back(Em);
    }
    return out;
}

// squared Sampson distance of a correspondence (normalised coordinates) to the epipolar constraint b^T E a = 0
inline double sampson2(const Mat3& E, const double a[2], const double b[2]) {
    const double Ea[3] = {E[0] * a[0] + E[1] * a[1] + E[2], E[3] * a[0] + E[4] * a[1] + E[5], E[6] * a[0] + E[7] * a[1] + E[8]};
    const double Etb[3] = {E[0] * b[0] + E[3] * b[1] + E[6], E[1] * b[0] + E[4] * b[1] + E[7], E[2] * b[0] + E[5] * b[1] + E[8]};
    const double r = b[0] * Ea[0] + b[1] * Ea[1] + Ea[2];
    const double d = Ea[0] * Ea[0] + Ea[1] * Ea[1] + Etb[0] * Etb[0] + Etb[1] * Etb[1];
    return d > 0. ? r * r / d : 1e300;
}

struct Motion {
    bool ok = false;
    Mat3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    double t[3]{0, 0, 1};  // unit length
    Mat3 E{};
    int inliers = 0, in_front = 0, samples = 0;
};

// The four decompositions of E and the one most inliers are in front of both cameras with (cv::recoverPose): x_b = R x_a + t.
inline void poseFromEssential(const Mat3& E, const std::vector<std::array<double, 2>>& a, const std::vector<std::array<double, 2>>& b,
                              const std::vector<char>& inlier, Motion& m) {
    using namespace detail;
    // SVD of E through the symmetric eigenproblem of E^T E: E = U diag(s, s, 0) V^T
    std::array<double, 9> EtE{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) EtE[3 * i + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
    std::array<double, 3> w;
    std::array<double, 9> Vc;
    symmetric_eigen<3>(EtE, w, Vc);  // ascending: column 0 = null direction of E
    // V = [v2 v1 v0] (descending singular values), U columns u_i = E v_i / s_i for the two non-zero ones, u3 = u1 x u2
    double V[3][3], U[3][3];
    for (int r = 0; r < 3; ++r) {
        V[r][0] = Vc[r * 3 + 2];
        V[r][1] = Vc[r * 3 + 1];
        V[r][2] = Vc[r * 3 + 0];
    }
    for (int c = 0; c < 2; ++c) {
        double u[3] = {E[0] * V[0][c] + E[1] * V[1][c] + E[2] * V[2][c], E[3] * V[0][c] + E[4] * V[1][c] + E[5] * V[2][c],
                       E[6] * V[0][c] + E[7] * V[1][c] + E[8] * V[2][c]};
        const double n = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int r = 0; r < 3; ++r) U[r][c] = n > 0. ? u[r] / n : (r == c ? 1. : 0.);
    }
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    {   // right-handed V as well (the sign of its last column is free)
        const double d = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                         V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
        if (d < 0.)
            for (int r = 0; r < 3; ++r) V[r][2] = -V[r][2];
    }
    const double Wm[2][9] = {{0, -1, 0, 1, 0, 0, 0, 0, 1}, {0, 1, 0, -1, 0, 0, 0, 0, 1}};
    int best = -1;
    for (int cand = 0; cand < 4; ++cand) {
        Mat3 Um, Vt, Wc;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                Um[3 * r + c] = U[r][c];
                Vt[3 * r + c] = V[c][r];
                Wc[3 * r + c] = Wm[cand & 1][3 * r + c];
            }
        Mat3 R = mul3(mul3(Um, Wc), Vt);
        if (det3(R) < 0.)
            for (double& e : R) e = -e;
        const double sgn = (cand & 2) ? -1. : 1.;
        const double t[3] = {sgn * U[0][2], sgn * U[1][2], sgn * U[2][2]};
        // depth of every inlier in both cameras: lambda_a (R xa) + t = lambda_b xb, least squares in (lambda_a, lambda_b)
        int front = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            if (!inlier[i]) continue;
            const double xa[3] = {a[i][0], a[i][1], 1.}, xb[3] = {b[i][0], b[i][1], 1.};
            const double ra[3] = {R[0] * xa[0] + R[1] * xa[1] + R[2] * xa[2], R[3] * xa[0] + R[4] * xa[1] + R[5] * xa[2],
                                  R[6] * xa[0] + R[7] * xa[1] + R[8] * xa[2]};
            // [ra -xb] [la lb]^T = -t
            const double a11 = ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2], a12 = -(ra[0] * xb[0] + ra[1] * xb[1] + ra[2] * xb[2]),
                         a22 = xb[0] * xb[0] + xb[1] * xb[1] + xb[2] * xb[2];
            const double b1 = -(ra[0] * t[0] + ra[1] * t[1] + ra[2] * t[2]), b2 = xb[0] * t[0] + xb[1] * t[1] + xb[2] * t[2];
            const double det = a11 * a22 - a12 * a12;
            if (std::fabs(det) < 1e-18) continue;
            const double la = (b1 * a22 - a12 * b2) / det, lb = (a11 * b2 - a12 * b1) / det;
            if (la > 0. && lb > 0. && la < 1e4 && lb < 1e4) ++front;  // (cv::recoverPose drops points farther than 50 baselines by default; kept generous here)
        }
        if (front > best) {
            best = front;
            m.R = R;
            for (int r = 0; r < 3; ++r) m.t[r] = t[r];
            m.in_front = front;
        }
    }
}

// cv::findEssentialMat(a, b, focal, pp, RANSAC, probability, threshold_px) + cv::recoverPose(E, a, b): pixel coordinates in,
// x_b = R x_a + t out.  ok = false with fewer than five correspondences or when no sample gave a model.
inline Motion estimateMotion(const std::vector<Vector2d>& pts_a, const std::vector<Vector2d>& pts_b, double focal, const Vector2d& pp,
                             double probability = 0.999, double threshold_px = 2.0, uint64_t seed = 1, int max_samples = 1000) {
    Motion m;
    const size_t n = std::min(pts_a.size(), pts_b.size());
    if (n < 5) return m;
    std::vector<std::array<double, 2>> a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
        a[i] = {(pts_a[i][0] - pp[0]) / focal, (pts_a[i][1] - pp[1]) / focal};
        b[i] = {(pts_b[i][0] - pp[0]) / focal, (pts_b[i][1] - pp[1]) / focal};
    }
    const double thr2 = (threshold_px / focal) * (threshold_px / focal);
    uint64_t state = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next = [&state]() {  // splitmix64
        uint64_t x = (state += 0x9E3779B97F4A7C15ull);
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    int need = max_samples;
    std::vector<char> best_inl(n, 0);
    for (int s = 0; s < need && s < max_samples; ++s) {
        size_t pick[5];
        for (int q = 0; q < 5; ++q) {
            bool fresh;
            do {
                pick[q] = (size_t)(next() % n);
                fresh = true;
                for (int r = 0; r < q; ++r) fresh = fresh && pick[r] != pick[q];
            } while (!fresh);
        }
        double sa[5][2], sb[5][2];
        for (int q = 0; q < 5; ++q) {
            sa[q][0] = a[pick[q]][0];
            sa[q][1] = a[pick[q]][1];
            sb[q][0] = b[pick[q]][0];
            sb[q][1] = b[pick[q]][1];
        }
        ++m.samples;
        for (const Mat3& E : essentialFromFive(sa, sb)) {
            // inlier count; abandoned as soon as even all remaining points could not beat the best model so far
            int count = 0;
            for (size_t i = 0; i < n; ++i) {
                count += sampson2(E, a[i].data(), b[i].data()) < thr2;
                if (count + (int)(n - 1 - i) <= m.inliers) break;
            }
            if (count > m.inliers) {
                m.inliers = count;
                m.E = E;
                m.ok = true;
                // samples needed for `probability` of one all-inlier sample at this inlier ratio (cv::RANSACUpdateNumIters)
                const double wr = std::min(1. - 1e-12, std::pow((double)count / (double)n, 5.));
                const double k = std::log(std::max(1e-300, 1. - probability)) / std::log(std::max(1e-300, 1. - wr));
                need = (int)std::min<double>(max_samples, std::max(1., std::ceil(k)));
            }
        }
    }
    if (m.ok) {
        for (size_t i = 0; i < n; ++i) best_inl[i] = sampson2(m.E, a[i].data(), b[i].data()) < thr2;
        poseFromEssential(m.E, a, b, best_inl, m);
    }
    return m;
}

// helpers::getMeanFlow (general_helpers.hpp:77-92)
inline double meanFlow(const std::vector<Vector2d>& p0, const std::vector<Vector2d>& p1) {
    if (p0.empty() || p0.size() != p1.size()) return 0.;
    double s = 0.;
    for (size_t i = 0; i < p0.size(); ++i) s += std::sqrt((p0[i][0] - p1[i][0]) * (p0[i][0] - p1[i][0]) + (p0[i][1] - p1[i][1]) * (p0[i][1] - p1[i][1]));
    return s / (double)p0.size();
}

// helpers::getMatches (general_helpers.hpp:35-74): the tracks that have a point at both stamps and no outlier label
inline void matches(const Tracklets& tracklets, TimestampNSec stamp_last, TimestampNSec stamp_cur, const std::set<int>& outlier_labels,
                    std::vector<Vector2d>& last_points, std::vector<Vector2d>& cur_points) {
    last_points.clear();
    cur_points.clear();
    int i_last = -1, i_cur = -1;
    for (size_t i = 0; i < tracklets.stamps.size(); ++i) {
        if (tracklets.stamps[i] == stamp_last) i_last = (int)i;
        if (tracklets.stamps[i] == stamp_cur) i_cur = (int)i;
    }
    if (i_last < 0 || i_cur < 0) return;
    for (const auto& track : tracklets.tracks) {
        if ((int)track.feature_points.size() > i_last && (int)track.feature_points.size() > i_cur && !outlier_labels.count(track.label)) {
            last_points.push_back(Vector2d(track.feature_points[i_last].u, track.feature_points[i_last].v));
            cur_points.push_back(Vector2d(track.feature_points[i_cur].u, track.feature_points[i_cur].v));
        }
    }
}

// helpers::getMotionUnscaled (general_helpers.hpp:209-231): motion of the VEHICLE from the last keyframe (t0) to the current frame
// (t1), new vehicle <- old vehicle, with |translation| = speed x dt; the five-point direction where the image flow allows it
// (mean flow >= 5 px), straight ahead along the camera's z axis otherwise.
inline EigenPose motionUnscaled(double focal, const Vector2d& pp, TimestampNSec stamp_cur, TimestampNSec stamp_last_kf, const Tracklets& tracklets,
                                const EigenPose& T_camera_vehicle, double speed_m_per_second, uint64_t seed = 1, Motion* info = nullptr) {
    std::vector<Vector2d> last_points, cur_points;
    matches(tracklets, stamp_last_kf, stamp_cur, {23, 24, 25, 26}, last_points, cur_points);
    EigenPose cam_t0_t1 = EigenPose::Identity();  // camera at t0 <- camera at t1
    cam_t0_t1.t[2] = 1.;
    if (!last_points.empty() && meanFlow(last_points, cur_points) >= 5.) {
        // calcMotion5Point: findEssentialMat(points1 = cur, points0 = last), recoverPose(E, cur, last): x_last = R x_cur + t
        const Motion m = estimateMotion(cur_points, last_points, focal, pp, 0.999, 2.0, seed);
        if (info) *info = m;
        if (m.ok) {
            for (int i = 0; i < 9; ++i) cam_t0_t1.R[i] = m.R[i];
            for (int i = 0; i < 3; ++i) cam_t0_t1.t[i] = m.t[i];
        }
    } else if (!last_points.empty()) {
        cam_t0_t1.t[2] = 0.;  // "not enough flow": calcMotion5Point zeroes the translation before it returns (:118-121)
    }
    const double dt = convert(stamp_cur) - convert(stamp_last_kf);
    const double n = std::sqrt(cam_t0_t1.t[0] * cam_t0_t1.t[0] + cam_t0_t1.t[1] * cam_t0_t1.t[1] + cam_t0_t1.t[2] * cam_t0_t1.t[2]);
    for (int i = 0; i < 3; ++i) cam_t0_t1.t[i] = cam_t0_t1.t[i] / std::max(0.0001, n) * speed_m_per_second * dt;
    return T_camera_vehicle.inverse() * cam_t0_t1.inverse() * T_camera_vehicle;
}

}  // namespace five_point
}  // namespace keyframe_bundle_adjustment
