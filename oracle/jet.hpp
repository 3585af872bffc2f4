// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it.
//
// jet.hpp — forward-mode dual numbers, the arithmetic Ceres' AutoDiffCostFunction runs the reference's
// templated functors with (ceres::Jet<double,N>; third-party, Ceres 1.13, not in the reference tree).
// Restated from the published definition: f(a + v·eps) = f(a) + f'(a) v·eps.
#pragma once
#include <cmath>

namespace kba_oracle {

template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) {
        for (int i = 0; i < N; ++i) v[i] = 0.0;
    }
    Jet(double s) : a(s) {  // NOLINT implicit, like ceres::Jet
        for (int i = 0; i < N; ++i) v[i] = 0.0;
    }
    Jet(double s, int k) : a(s) {
        for (int i = 0; i < N; ++i) v[i] = 0.0;
        v[k] = 1.0;
    }
    Jet& operator+=(const Jet& o) {
        a += o.a;
        for (int i = 0; i < N; ++i) v[i] += o.v[i];
        return *this;
    }
    Jet& operator-=(const Jet& o) {
        a -= o.a;
        for (int i = 0; i < N; ++i) v[i] -= o.v[i];
        return *this;
    }
    Jet& operator*=(const Jet& o) {
        *this = *this * o;
        return *this;
    }
    Jet& operator/=(const Jet& o) {
        *this = *this / o;
        return *this;
    }
};

template <int N>
inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h;
    h.a = f.a + g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
    return h;
}
template <int N>
inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h;
    h.a = f.a - g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
    return h;
}
template <int N>
inline Jet<N> operator-(const Jet<N>& f) {
    Jet<N> h;
    h.a = -f.a;
    for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
    return h;
}
template <int N>
inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h;
    h.a = f.a * g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
template <int N>
inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // (f/g)' = (f' - f/g g') / g
    Jet<N> h;
    const double g_inv = 1.0 / g.a;
    const double fg = f.a * g_inv;
    h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * g_inv;
    return h;
}
// mixed with double
template <int N>
inline Jet<N> operator+(const Jet<N>& f, double s) {
    Jet<N> h = f;
    h.a += s;
    return h;
}
template <int N>
inline Jet<N> operator+(double s, const Jet<N>& f) {
    return f + s;
}
template <int N>
inline Jet<N> operator-(const Jet<N>& f, double s) {
    Jet<N> h = f;
    h.a -= s;
    return h;
}
template <int N>
inline Jet<N> operator-(double s, const Jet<N>& f) {
    Jet<N> h = -f;
    h.a += s;
    return h;
}
template <int N>
inline Jet<N> operator*(const Jet<N>& f, double s) {
    Jet<N> h;
    h.a = f.a * s;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s;
    return h;
}
template <int N>
inline Jet<N> operator*(double s, const Jet<N>& f) {
    return f * s;
}
template <int N>
inline Jet<N> operator/(const Jet<N>& f, double s) {
    return f * (1.0 / s);
}
template <int N>
inline Jet<N> operator/(double s, const Jet<N>& g) {
    Jet<N> h;
    const double g_inv = 1.0 / g.a;
    h.a = s * g_inv;
    const double m = -s * g_inv * g_inv;
    for (int i = 0; i < N; ++i) h.v[i] = m * g.v[i];
    return h;
}
// comparisons act on the scalar part (as ceres::Jet)
template <int N>
inline bool operator<(const Jet<N>& f, const Jet<N>& g) {
    return f.a < g.a;
}
template <int N>
inline bool operator>=(const Jet<N>& f, const Jet<N>& g) {
    return f.a >= g.a;
}
template <int N>
inline bool operator>(const Jet<N>& f, const Jet<N>& g) {
    return f.a > g.a;
}

inline double jsqrt(double x) {
    return std::sqrt(x);
}
inline double jabs(double x) {
    return std::fabs(x);
}
template <int N>
inline Jet<N> jsqrt(const Jet<N>& f) {
    Jet<N> h;
    h.a = std::sqrt(f.a);
    const double m = 1.0 / (2.0 * h.a);
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * m;
    return h;
}
template <int N>
inline Jet<N> jabs(const Jet<N>& f) {
    return f.a < 0.0 ? -f : f;
}
inline double scalar_of(double x) {
    return x;
}
template <int N>
inline double scalar_of(const Jet<N>& f) {
    return f.a;
}

}  // namespace kba_oracle
