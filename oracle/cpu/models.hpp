// AD scalar, the three reference models and the RK4 discrete map of the CPU oracle port.
// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); included by mpc_oracle.cpp.
#pragma once
// Problem data: cartpole rlmpc/mpc/cartpole/acados.py:71-92 + config/cartpole.yaml; linear rlmpc/mpc/linear_system/acados.py:27-131;
// chain rlmpc/mpc/chain_mass/ocp_utils.py:42-147,253-277.
#include "mpc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace oracle_impl {

constexpr double INF_BOUND = 1e29;
constexpr int IPM_MAX_ITER = 60;
constexpr double IPM_TOL_RES = 1e-9, IPM_TOL_MU = 1e-11, IPM_T_MIN = 1e-1, IPM_MU0 = 1.0, IPM_FRAC = 0.995;
constexpr double IPM_WARM_C = 1e-4, IPM_WARM_MIN = 1e-10, IPM_WARM_MAX = 3e-2;   // warm start: mu_w = clamp(C step^2)
// inexact SQP: QP tolerances follow the NLP residual r: tol_res = clamp(C r^2, TOL_RES, CAP), tol_mu = clamp(C r^2 / 100, TOL_MU, CAP / 10);
// convergence is only declared after a QP solved to the tight tolerances
constexpr double IPM_ADAPT_C = 1e1, IPM_ADAPT_CAP = 3e-2;
constexpr double IPM_SKIP_SIGMA = 3e-6;   // oracle/sqp_dense.py: the predictor step is taken where it alone cuts the complementarity this far
// Adjoint (sensitivity) solve: the stiffness lam / t of an active bound row is capped.  A row whose slack the interior point took to
// 1e-18 pins its coordinate either way (the answer moves by O(1 / stiffness)), but 1e19 on the diagonal of a STATE block costs the Riccati
// recursion all sixteen digits of the entries next to it.  Measured against the dense solve of the mirror on the hardest instances of
// tests/test_gpu_fullsize.py: cap 1e8 -> 2e-7, 1e9 -> 1e-7 (bias 1e-8), 1e10 -> 1e-6, 1e12 -> 9e-5, 1e14 -> 3e-2, none -> 5e-1.
constexpr double SENS_W_MAX = 1e9;

// ------------------------------------------------------------------------------------------------
// forward-mode AD scalar, nestable
// ------------------------------------------------------------------------------------------------
template <class T, int N>
struct Dual {
    T v;
    T d[N];
    Dual() {}
    Dual(double c) : v(c) {
        for (int i = 0; i < N; ++i) d[i] = T(0.0);
    }
    friend Dual operator+(const Dual &a, const Dual &b) {
        Dual r;
        r.v = a.v + b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
        return r;
    }
    friend Dual operator-(const Dual &a, const Dual &b) {
        Dual r;
        r.v = a.v - b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
        return r;
    }
    friend Dual operator-(const Dual &a) {
        Dual r;
        r.v = -a.v;
        for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
        return r;
    }
    friend Dual operator*(const Dual &a, const Dual &b) {
        Dual r;
        r.v = a.v * b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
        return r;
    }
    friend Dual operator/(const Dual &a, const Dual &b) {
        Dual r;
        T inv = T(1.0) / b.v;
        r.v = a.v * inv;
        for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
        return r;
    }
};
inline double ad_sin(double x) { return std::sin(x); }
inline double ad_cos(double x) { return std::cos(x); }
inline double ad_sqrt(double x) { return std::sqrt(x); }
template <class T, int N>
Dual<T, N> ad_sin(const Dual<T, N> &a) {
    Dual<T, N> r;
    r.v = ad_sin(a.v);
    T c = ad_cos(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class T, int N>
Dual<T, N> ad_cos(const Dual<T, N> &a) {
    Dual<T, N> r;
    r.v = ad_cos(a.v);
    T s = T(0.0) - ad_sin(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
    return r;
}
template <class T, int N>
Dual<T, N> ad_sqrt(const Dual<T, N> &a) {
    Dual<T, N> r;
    r.v = ad_sqrt(a.v);
    T h = T(0.5) / r.v;
    for (int i = 0; i < N; ++i) r.d[i] = h * a.d[i];
    return r;
}
inline double val(double x) { return x; }
template <class T, int N>
double val(const Dual<T, N> &a) { return val(a.v); }

// ------------------------------------------------------------------------------------------------
// models.  Interface:
//   NX, NU, NP, NTD (number of parameters the DYNAMICS depend on), td_index(i) -> index into p
//   ode<S>(x,u,th,f,sp)        continuous dynamics (th = the NTD dynamics parameters)   [linear: discrete map]
//   cost_val/grad/hess/dp/mixed   stage cost l_k (unscaled), stage vector order v = [u; x], k == N terminal
// ------------------------------------------------------------------------------------------------
struct Cartpole {
    static constexpr int NX = 4, NU = 1, NP = 83, NTD = 3;
    static constexpr bool DISCRETE = false, SKIP_CORRECTOR = true;
    static constexpr bool SENS_EXTRAP = true;    // adjoint solve of du0/dp: Richardson extrapolation in the stiffness cap where a STATE row is capped (mpc_oracle.cpp)
    static constexpr double TOL_MU_FACTOR = 0.1;
    static int td_index(int i) { return i; }
    // cost block of the full parameter vector p (nlp.py:969-989, column-major): W_0 (5x5) at 3, W at 28, W_e (4x4) at 53, yref_0 at 69,
    // yref at 74, yref_e at 79.  The solver sees whatever set_parameter / cost_set wrote there (mpc.py:233-257); the mirror's cost
    // is NOT parameterised by them (nlp.py:1039-1055), so their gradient entries stay zero (cost_dp / cost_mixed below).
    static constexpr int P_W0 = 3, P_W = 28, P_WE = 53, P_YREF0 = 69, P_YREF = 74, P_YREFE = 79;
    static const double *Wof(int k, const double *p) { return p + (k == 0 ? P_W0 : P_W); }
    static const double *yrof(int k, const double *p) { return p + (k == 0 ? P_YREF0 : P_YREF); }
    template <class S>
    static void ode(const S *x, const S *u, const S *th, S *f, const OracleSpec &) {
        const double g = 9.8;
        S M = th[0], m = th[1], l = th[2];
        S c = ad_cos(x[2]), s = ad_sin(x[2]);
        S temp = (u[0] + m * x[3] * x[3] * s) / (m + M);
        S thdd = (S(g) * s - c * temp) / (l * (S(4.0 / 3.0) - m * c * c / (m + M)));
        f[0] = x[1];
        f[1] = temp - m * thdd * c / (m + M);
        f[2] = x[3];
        f[3] = thdd;
    }
    // y index of stage-vector coordinate i (v = [u; x] -> y = [x; u])
    static int yi(int i) { return i < NU ? NX + i : i - NU; }
    static double cost_val(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &) {
        const double *W = Wof(k, p), *yr = yrof(k, p), *We = p + P_WE, *yre = p + P_YREFE;
        double v = 0;
        if (k < N) {
            double y[5] = {x[0] - yr[0], x[1] - yr[1], x[2] - yr[2], x[3] - yr[3], u[0] - yr[4]};
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 5; ++j) v += 0.5 * y[i] * W[j * 5 + i] * y[j];
        } else {
            double y[4] = {x[0] - yre[0], x[1] - yre[1], x[2] - yre[2], x[3] - yre[3]};
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) v += 0.5 * y[i] * We[j * 4 + i] * y[j];
        }
        return v;
    }
    static void cost_grad(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &, double *g) {
        const double *W = Wof(k, p), *yr = yrof(k, p), *We = p + P_WE, *yre = p + P_YREFE;
        if (k < N) {
            double y[5] = {x[0] - yr[0], x[1] - yr[1], x[2] - yr[2], x[3] - yr[3], u[0] - yr[4]};
            for (int i = 0; i < 5; ++i) {
                double a = 0;
                for (int j = 0; j < 5; ++j) a += 0.5 * (W[j * 5 + yi(i)] + W[yi(i) * 5 + j]) * y[j];
                g[i] = a;
            }
        } else {
            double y[4] = {x[0] - yre[0], x[1] - yre[1], x[2] - yre[2], x[3] - yre[3]};
            g[0] = 0;
            for (int i = 0; i < 4; ++i) {
                double a = 0;
                for (int j = 0; j < 4; ++j) a += 0.5 * (We[j * 4 + i] + We[i * 4 + j]) * y[j];
                g[NU + i] = a;
            }
        }
    }
    static void cost_hess(int k, int N, const double *, const double *, const double *p, const OracleSpec &, double *H) {
        const double *W = Wof(k, p), *We = p + P_WE;
        constexpr int NW = NX + NU;
        for (int i = 0; i < NW * NW; ++i) H[i] = 0;
        if (k < N) {
            for (int i = 0; i < NW; ++i)
                for (int j = 0; j < NW; ++j) H[i * NW + j] = 0.5 * (W[yi(j) * 5 + yi(i)] + W[yi(i) * 5 + yi(j)]);
        } else {
            for (int i = 0; i < NX; ++i)
                for (int j = 0; j < NX; ++j) H[(NU + i) * NW + NU + j] = 0.5 * (We[j * 4 + i] + We[i * 4 + j]);
        }
    }
    // non-parameterised NLS mirror: the cost does not depend on p (nlp.py:1039-1055)
    static void cost_dp(int, int, const double *, const double *, const double *, const OracleSpec &, double, double *) {}
    static void cost_mixed(int, int, const double *, const double *, const double *, const OracleSpec &, const double *, double,
                           double *) {}
};

struct Linear {
    static constexpr int NX = 2, NU = 1, NP = 12, NTD = 8;
    static constexpr bool DISCRETE = true, SKIP_CORRECTOR = false;
    static constexpr bool SENS_EXTRAP = false;
    static constexpr double TOL_MU_FACTOR = 0.1;
    static int td_index(int i) { return i; }
    // consts: P (2x2 row-major) terminal DARE matrix
    template <class S>
    static void ode(const S *x, const S *u, const S *th, S *f, const OracleSpec &) {
        // discrete map x+ = A x + B u + b, A column-major in p (acados.py:60-62,89-90)
        f[0] = th[0] * x[0] + th[2] * x[1] + th[4] * u[0] + th[6];
        f[1] = th[1] * x[0] + th[3] * x[1] + th[5] * u[0] + th[7];
    }
    static double cost_val(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &sp) {
        if (k < N) {
            double v = 0.5 * (x[0] * x[0] + x[1] * x[1] + u[0] * u[0]) + p[9] * x[0] + p[10] * x[1] + p[11] * u[0];
            return k == 0 ? v + p[8] : v;
        }
        const double *P = sp.consts;
        return 0.5 * (x[0] * (P[0] * x[0] + P[1] * x[1]) + x[1] * (P[2] * x[0] + P[3] * x[1]));
    }
    static void cost_grad(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &sp, double *g) {
        if (k < N) {
            g[0] = u[0] + p[11];
            g[1] = x[0] + p[9];
            g[2] = x[1] + p[10];
        } else {
            const double *P = sp.consts;
            g[0] = 0;
            g[1] = P[0] * x[0] + 0.5 * (P[1] + P[2]) * x[1];
            g[2] = 0.5 * (P[1] + P[2]) * x[0] + P[3] * x[1];
        }
    }
    static void cost_hess(int k, int N, const double *, const double *, const double *, const OracleSpec &sp, double *H) {
        for (int i = 0; i < 9; ++i) H[i] = 0;
        if (k < N) {
            H[0] = H[4] = H[8] = 1.0;
        } else {
            const double *P = sp.consts;
            H[4] = P[0];
            H[5] = H[7] = 0.5 * (P[1] + P[2]);
            H[8] = P[3];
        }
    }
    static void cost_dp(int k, int N, const double *x, const double *u, const double *, const OracleSpec &, double sc, double *out) {
        if (k < N) {
            out[9] += sc * x[0];
            out[10] += sc * x[1];
            out[11] += sc * u[0];
            if (k == 0) out[8] += sc;
        }
    }
    static void cost_mixed(int k, int N, const double *, const double *, const double *, const OracleSpec &, const double *y,
                           double sc, double *out) {
        if (k < N) {   // d2 l / dv df = I (y order [x;u], v order [u;x])
            out[9] += sc * y[1];
            out[10] += sc * y[2];
            out[11] += sc * y[0];
        }
    }
};

template <int NMASS>
struct Chain {
    static constexpr int M = NMASS - 2, NL = NMASS - 1;
    static constexpr int NX = (2 * M + 1) * 3, NU = 3;
    static constexpr int OFF_M = 0, OFF_D = NL, OFF_L = 4 * NL, OFF_C = 7 * NL, OFF_Q = 10 * NL, OFF_R = OFF_Q + NX * NX,
                         OFF_W = OFF_R + NU * NU;
    static constexpr int NP = OFF_W + 3 * M, NTD = 10 * NL + 3 * M;
    static constexpr bool DISCRETE = false, SKIP_CORRECTOR = false;
    static constexpr bool SENS_EXTRAP = false;
    // complementarity tolerance of an inexact QP = TOL_MU_FACTOR x its residual tolerance (1/10 elsewhere): with the cap itself the
    // n_mass 7 chain needs 15.8 instead of 18.8 interior-point iterations per solve, n_mass 3 / 5 and every SQP iteration count unchanged
    static constexpr double TOL_MU_FACTOR = 1.0;
    static int td_index(int i) { return i < 10 * NL ? i : OFF_W + (i - 10 * NL); }
    // consts: x_ss (NX)
    template <class S>
    static void ode(const S *x, const S *u, const S *th, S *f, const OracleSpec &) {
        const S *pos = x, *vel = x + 3 * (M + 1);
        const S *m = th, *D = th + NL, *L = th + 4 * NL, *C = th + 7 * NL, *w = th + 10 * NL;
        S acc[3 * M];
        for (int i = 0; i < M; ++i) {
            acc[3 * i] = w[3 * i];
            acc[3 * i + 1] = w[3 * i + 1];
            acc[3 * i + 2] = w[3 * i + 2] - S(9.81);
        }
        for (int i = 0; i <= M; ++i) {
            S dist[3], n2 = S(0.0);
            for (int j = 0; j < 3; ++j) {
                dist[j] = i ? pos[3 * i + j] - pos[3 * (i - 1) + j] : pos[j];
                n2 = n2 + dist[j] * dist[j];
            }
            S nrm = ad_sqrt(n2);
            for (int j = 0; j < 3; ++j) {
                S Fs = D[3 * i + j] / m[i] * (S(1.0) - L[3 * i + j] / nrm) * dist[j];
                S vr = i < M ? vel[3 * i + j] : u[j];
                S dv = i ? vr - vel[3 * (i - 1) + j] : vr;
                S Ft = Fs + C[3 * i + j] * dv;
                if (i < M) acc[3 * i + j] = acc[3 * i + j] - Ft;
                if (i > 0) acc[3 * (i - 1) + j] = acc[3 * (i - 1) + j] + Ft;
            }
        }
        for (int i = 0; i < 3 * M; ++i) f[i] = vel[i];
        for (int j = 0; j < 3; ++j) f[3 * M + j] = u[j];
        for (int i = 0; i < 3 * M; ++i) f[3 * (M + 1) + i] = acc[i];
    }
    // Q, R column-major in p (ocp_utils.py:267,273): Q(i,j) = p[OFF_Q + j*NX + i]
    static double Qe(const double *p, int i, int j) { return p[OFF_Q + j * NX + i]; }
    static double Re(const double *p, int i, int j) { return p[OFF_R + j * NU + i]; }
    static double cost_val(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &sp) {
        const double *xs = sp.consts;
        double v = 0;
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) v += 0.5 * (x[i] - xs[i]) * Qe(p, i, j) * (x[j] - xs[j]);
        if (k < N)
            for (int i = 0; i < NU; ++i)
                for (int j = 0; j < NU; ++j) v += 0.5 * u[i] * Re(p, i, j) * u[j];
        return v;
    }
    static void cost_grad(int k, int N, const double *x, const double *u, const double *p, const OracleSpec &sp, double *g) {
        const double *xs = sp.consts;
        for (int i = 0; i < NU; ++i) {
            double a = 0;
            if (k < N)
                for (int j = 0; j < NU; ++j) a += 0.5 * (Re(p, i, j) + Re(p, j, i)) * u[j];
            g[i] = a;
        }
        for (int i = 0; i < NX; ++i) {
            double a = 0;
            for (int j = 0; j < NX; ++j) a += 0.5 * (Qe(p, i, j) + Qe(p, j, i)) * (x[j] - xs[j]);
            g[NU + i] = a;
        }
    }
    static void cost_hess(int k, int N, const double *, const double *, const double *p, const OracleSpec &, double *H) {
        constexpr int NW = NX + NU;
        for (int i = 0; i < NW * NW; ++i) H[i] = 0;
        if (k < N)
            for (int i = 0; i < NU; ++i)
                for (int j = 0; j < NU; ++j) H[i * NW + j] = 0.5 * (Re(p, i, j) + Re(p, j, i));
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) H[(NU + i) * NW + NU + j] = 0.5 * (Qe(p, i, j) + Qe(p, j, i));
    }
    static void cost_dp(int k, int N, const double *x, const double *u, const double *, const OracleSpec &sp, double sc, double *out) {
        const double *xs = sp.consts;
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) out[OFF_Q + j * NX + i] += sc * 0.5 * (x[i] - xs[i]) * (x[j] - xs[j]);
        if (k < N)
            for (int i = 0; i < NU; ++i)
                for (int j = 0; j < NU; ++j) out[OFF_R + j * NU + i] += sc * 0.5 * u[i] * u[j];
    }
    static void cost_mixed(int k, int N, const double *x, const double *u, const double *, const OracleSpec &sp, const double *y,
                           double sc, double *out) {
        // d/dQ_ij of y_x' (1/2 (Q+Q') e) = 1/2 (y_i e_j + y_j e_i)
        const double *xs = sp.consts;
        const double *yu = y, *yx = y + NU;
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j) out[OFF_Q + j * NX + i] += sc * 0.5 * (yx[i] * (x[j] - xs[j]) + yx[j] * (x[i] - xs[i]));
        if (k < N)
            for (int i = 0; i < NU; ++i)
                for (int j = 0; j < NU; ++j) out[OFF_R + j * NU + i] += sc * 0.5 * (yu[i] * u[j] + yu[j] * u[i]);
    }
};

// ------------------------------------------------------------------------------------------------
// discrete map F = RK4^rk_steps(ode; h)  (common/integrator.py:6-33, chain_mass/ocp_utils.py:42-56)
// ------------------------------------------------------------------------------------------------
template <class Mdl, class S>
void disc_map(const S *x, const S *u, const S *th, S *xn, const OracleSpec &sp) {
    constexpr int NX = Mdl::NX;
    if (Mdl::DISCRETE) {
        Mdl::template ode<S>(x, u, th, xn, sp);
        return;
    }
    S xc[NX], k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
    for (int i = 0; i < NX; ++i) xc[i] = x[i];
    const double h = sp.h;
    for (int s = 0; s < sp.rk_steps; ++s) {
        Mdl::template ode<S>(xc, u, th, k1, sp);
        for (int i = 0; i < NX; ++i) xt[i] = xc[i] + S(h / 2) * k1[i];
        Mdl::template ode<S>(xt, u, th, k2, sp);
        for (int i = 0; i < NX; ++i) xt[i] = xc[i] + S(h / 2) * k2[i];
        Mdl::template ode<S>(xt, u, th, k3, sp);
        for (int i = 0; i < NX; ++i) xt[i] = xc[i] + S(h) * k3[i];
        Mdl::template ode<S>(xt, u, th, k4, sp);
        for (int i = 0; i < NX; ++i) xc[i] = xc[i] + S(h / 6) * (k1[i] + S(2.0) * k2[i] + S(2.0) * k3[i] + k4[i]);
    }
    for (int i = 0; i < NX; ++i) xn[i] = xc[i];
}

}  // namespace oracle_impl
