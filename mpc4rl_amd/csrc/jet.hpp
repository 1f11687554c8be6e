// Forward-mode derivative scalars for the device-side model code (gfx950, fp64).
//
//   Jet1<N> : value + N first-order tangents.                                      A_k, B_k, dF/dtheta
//   Jet2<N> : value, one "inner" tangent e (direction y), N "outer" tangents g_i,
//             and the mixed second-order terms m_i = d/d eps ( d f / d v_i ) along y.   Hessian-vector products
//
// Jet2 is the flattened form of a dual-of-dual; carrying only ONE inner direction keeps the
// register footprint at 2N+2 doubles per scalar, which is what lets a whole RK4 step of the stage
// dynamics stay in VGPRs of a single lane.  All loops are compile-time so nothing is runtime-indexed.
#pragma once
#include <hip/hip_runtime.h>

namespace mpcrl {

#define MPCRL_DI __device__ __forceinline__

// 1 / x from the hardware seed and two Newton steps (~1 ulp) for the denominators of the dynamics: an IEEE division is ~15 instructions
// of scaling, fix-up and special cases that a model's denominators (masses, lengths, norms) never need
MPCRL_DI double jet_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

template <int N>
struct Jet1 {
    double v;
    double d[N];
    MPCRL_DI Jet1() {}
    MPCRL_DI Jet1(double c) : v(c) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = 0.0;
    }
};

template <int N>
MPCRL_DI Jet1<N> operator+(const Jet1<N> &a, const Jet1<N> &b) {
    Jet1<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator-(const Jet1<N> &a, const Jet1<N> &b) {
    Jet1<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator*(const Jet1<N> &a, const Jet1<N> &b) {
    Jet1<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = fma(a.d[i], b.v, a.v * b.d[i]);
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator/(const Jet1<N> &a, const Jet1<N> &b) {
    Jet1<N> r;
    const double inv = jet_rcp(b.v);
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator*(double s, const Jet1<N> &a) {
    Jet1<N> r;
    r.v = s * a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator+(const Jet1<N> &a, double s) {
    Jet1<N> r = a;
    r.v += s;
    return r;
}
template <int N>
MPCRL_DI Jet1<N> operator-(double s, const Jet1<N> &a) {
    Jet1<N> r;
    r.v = s - a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
template <int N>
MPCRL_DI Jet1<N> jrecip(const Jet1<N> &b) {
    Jet1<N> r;
    r.v = jet_rcp(b.v);
    const double m2 = -r.v * r.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = m2 * b.d[i];
    return r;
}
template <int N>
MPCRL_DI void jsincos(const Jet1<N> &a, Jet1<N> &s, Jet1<N> &c) {
    double sv, cv;
    sincos(a.v, &sv, &cv);
    s.v = sv;
    c.v = cv;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.d[i] = cv * a.d[i];
        c.d[i] = -sv * a.d[i];
    }
}
template <int N>
MPCRL_DI Jet1<N> jsqrt(const Jet1<N> &a) {
    Jet1<N> r;
    r.v = sqrt(a.v);
    const double h = 0.5 / r.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = h * a.d[i];
    return r;
}

// ------------------------------------------------------------------------------------------------
template <int N>
struct Jet2 {
    double v, e;
    double g[N], m[N];
    MPCRL_DI Jet2() {}
    MPCRL_DI Jet2(double c) : v(c), e(0.0) {
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = 0.0, m[i] = 0.0;
    }
};
template <int N>
MPCRL_DI Jet2<N> operator+(const Jet2<N> &a, const Jet2<N> &b) {
    Jet2<N> r;
    r.v = a.v + b.v;
    r.e = a.e + b.e;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = a.g[i] + b.g[i], r.m[i] = a.m[i] + b.m[i];
    return r;
}
template <int N>
MPCRL_DI Jet2<N> operator-(const Jet2<N> &a, const Jet2<N> &b) {
    Jet2<N> r;
    r.v = a.v - b.v;
    r.e = a.e - b.e;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = a.g[i] - b.g[i], r.m[i] = a.m[i] - b.m[i];
    return r;
}
template <int N>
MPCRL_DI Jet2<N> operator*(const Jet2<N> &a, const Jet2<N> &b) {
    Jet2<N> r;
    r.v = a.v * b.v;
    r.e = fma(a.e, b.v, a.v * b.e);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.g[i] = fma(a.g[i], b.v, a.v * b.g[i]);
        r.m[i] = fma(a.m[i], b.v, fma(a.g[i], b.e, fma(a.e, b.g[i], a.v * b.m[i])));
    }
    return r;
}
template <int N>
MPCRL_DI Jet2<N> jrecip(const Jet2<N> &b) {
    Jet2<N> r;
    const double i1 = jet_rcp(b.v), i2 = i1 * i1, i3 = 2.0 * i2 * i1;
    r.v = i1;
    r.e = -b.e * i2;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.g[i] = -b.g[i] * i2;
        r.m[i] = fma(b.g[i] * b.e, i3, -b.m[i] * i2);
    }
    return r;
}
template <int N>
MPCRL_DI Jet2<N> operator/(const Jet2<N> &a, const Jet2<N> &b) {
    return a * jrecip(b);
}
template <int N>
MPCRL_DI Jet2<N> operator*(double s, const Jet2<N> &a) {
    Jet2<N> r;
    r.v = s * a.v;
    r.e = s * a.e;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = s * a.g[i], r.m[i] = s * a.m[i];
    return r;
}
template <int N>
MPCRL_DI Jet2<N> operator+(const Jet2<N> &a, double s) {
    Jet2<N> r = a;
    r.v += s;
    return r;
}
template <int N>
MPCRL_DI Jet2<N> operator-(double s, const Jet2<N> &a) {
    Jet2<N> r;
    r.v = s - a.v;
    r.e = -a.e;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = -a.g[i], r.m[i] = -a.m[i];
    return r;
}
template <int N>
MPCRL_DI void jsincos(const Jet2<N> &a, Jet2<N> &s, Jet2<N> &c) {
    double sv, cv;
    sincos(a.v, &sv, &cv);
    s.v = sv;
    c.v = cv;
    s.e = cv * a.e;
    c.e = -sv * a.e;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.g[i] = cv * a.g[i];
        c.g[i] = -sv * a.g[i];
        s.m[i] = fma(cv, a.m[i], -sv * a.e * a.g[i]);
        c.m[i] = -fma(sv, a.m[i], cv * a.e * a.g[i]);
    }
}
template <int N>
MPCRL_DI Jet2<N> jsqrt(const Jet2<N> &a) {
    Jet2<N> r;
    r.v = sqrt(a.v);
    const double h = 0.5 / r.v, h2 = -0.5 * h / a.v;
    r.e = h * a.e;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.g[i] = h * a.g[i];
        r.m[i] = fma(h, a.m[i], h2 * a.e * a.g[i]);
    }
    return r;
}

// ---- JetH<N>: value, gradient and the full symmetric Hessian along N directions (packed lower triangle, (i, j) at i (i + 1) / 2 + j).
// One evaluation of a map yields all its second derivatives along the directions; N evaluations with Jet2<N> (one second direction
// each) give the same numbers with every transcendental / reciprocal of the map computed N times.
template <int N>
struct JetH {
    static constexpr int NH = N * (N + 1) / 2;
    double v;
    double g[N], h[NH];
    MPCRL_DI JetH() {}
    MPCRL_DI JetH(double c) : v(c) {
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NH; ++i) h[i] = 0.0;
    }
};
// r = f(a) given f, f', f'' at a.v
template <int N>
MPCRL_DI JetH<N> jeth_chain(const JetH<N> &a, double f0, double f1, double f2) {
    JetH<N> r;
    r.v = f0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = f1 * a.g[i];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) r.h[i * (i + 1) / 2 + j] = fma(f1, a.h[i * (i + 1) / 2 + j], f2 * a.g[i] * a.g[j]);
    return r;
}
template <int N>
MPCRL_DI JetH<N> operator+(const JetH<N> &a, const JetH<N> &b) {
    JetH<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = a.g[i] + b.g[i];
#pragma unroll
    for (int i = 0; i < JetH<N>::NH; ++i) r.h[i] = a.h[i] + b.h[i];
    return r;
}
template <int N>
MPCRL_DI JetH<N> operator-(const JetH<N> &a, const JetH<N> &b) {
    JetH<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = a.g[i] - b.g[i];
#pragma unroll
    for (int i = 0; i < JetH<N>::NH; ++i) r.h[i] = a.h[i] - b.h[i];
    return r;
}
template <int N>
MPCRL_DI JetH<N> operator*(const JetH<N> &a, const JetH<N> &b) {
    JetH<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = fma(a.g[i], b.v, a.v * b.g[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const int e = i * (i + 1) / 2 + j;
            r.h[e] = fma(a.h[e], b.v, fma(a.g[i], b.g[j], fma(a.g[j], b.g[i], a.v * b.h[e])));
        }
    return r;
}
template <int N>
MPCRL_DI JetH<N> jrecip(const JetH<N> &b) {
    const double i1 = jet_rcp(b.v), i2 = i1 * i1;
    return jeth_chain(b, i1, -i2, 2.0 * i2 * i1);
}
template <int N>
MPCRL_DI JetH<N> operator/(const JetH<N> &a, const JetH<N> &b) {
    return a * jrecip(b);
}
template <int N>
MPCRL_DI JetH<N> operator*(double s, const JetH<N> &a) {
    JetH<N> r;
    r.v = s * a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = s * a.g[i];
#pragma unroll
    for (int i = 0; i < JetH<N>::NH; ++i) r.h[i] = s * a.h[i];
    return r;
}
template <int N>
MPCRL_DI JetH<N> operator+(const JetH<N> &a, double s) {
    JetH<N> r = a;
    r.v += s;
    return r;
}
template <int N>
MPCRL_DI JetH<N> operator-(double s, const JetH<N> &a) {
    JetH<N> r;
    r.v = s - a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.g[i] = -a.g[i];
#pragma unroll
    for (int i = 0; i < JetH<N>::NH; ++i) r.h[i] = -a.h[i];
    return r;
}
template <int N>
MPCRL_DI void jsincos(const JetH<N> &a, JetH<N> &s, JetH<N> &c) {
    double sv, cv;
    sincos(a.v, &sv, &cv);
    s = jeth_chain(a, sv, cv, -sv);
    c = jeth_chain(a, cv, -sv, -cv);
}
template <int N>
MPCRL_DI JetH<N> jsqrt(const JetH<N> &a) {
    const double r = sqrt(a.v), h = 0.5 / r;
    return jeth_chain(a, r, h, -0.5 * h / a.v);
}

// plain double overloads so model code can be instantiated for values only
MPCRL_DI void jsincos(const double &a, double &s, double &c) { sincos(a, &s, &c); }
MPCRL_DI double jsqrt(double a) { return sqrt(a); }
MPCRL_DI double jrecip(double a) { return 1.0 / a; }

}  // namespace mpcrl
