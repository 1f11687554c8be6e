// Device-side problem descriptors of the small (register-resident) OCPs.
//
// Each model states the reference's problem data as code:
//   Cartpole  rlmpc/mpc/cartpole/acados.py:71-92 (dynamics as written there: no `l` in temp / x_ddot),
//             cost y = [x; u], NONLINEAR_LS with numeric W, yref (config/cartpole.yaml:25-51)
//   Linear    rlmpc/mpc/linear_system/acados.py:27-70 (x+ = A x + B u + b, l = 1/2 y'y + f'y, l_0 = l + V_0,
//             l_e = 1/2 x' P x with P = DARE(A,B,I,I) numeric)
//
// Interface used by small_kernel.hip (stage vector order v = [u; x]):
//   NX, NU, NP          dims, length of the full parameter vector p (reference order, nlp.py:969-989)
//   NTD, td_index(i)    parameters the DYNAMICS depend on, and where they sit in p
//   NTC, tc_index(i)    parameters only the COST depends on
//   ode<S>(x,u,th,f)    continuous dynamics (or the discrete map when DISCRETE) for S = double | Jet1 | Jet2
//   cost_val / cost_grad / hess / cost_dp / cost_mixed    UNSCALED stage cost l_k and its derivatives
#pragma once
#include "jet.hpp"

namespace mpcrl {

constexpr int SMALL_MAXNW = 8;
constexpr int SMALL_MAXC = 64;

// Kernel-argument copy of MpcrlProblemSpec (include/mpcrl.h) for the small kernel.  Passed by value, so every
// constant below is a scalar (SGPR) operand in the kernel.
struct SmallSpec {
    int N, np, cost_kind, rk_steps, max_iter;
    int exit_window;      // mpcrl_set_exit_rule: check the progress every exit_window SQP iterations (0 = never: the reference's behaviour)
    double exit_factor;   // ... the best NLP residual so far must be below exit_factor x its value at the previous check
    double dT, gamma, h, tol;
    double lb0[SMALL_MAXNW], ub0[SMALL_MAXNW];   // nu used
    double lb[SMALL_MAXNW], ub[SMALL_MAXNW];     // nu + nx used, v = [u; x]
    double lbe[SMALL_MAXNW], ube[SMALL_MAXNW];   // nx used
    double zl[SMALL_MAXNW], zu[SMALL_MAXNW];
    int soft[SMALL_MAXNW];
    double consts[SMALL_MAXC];
};

struct CartpoleDev {
    static constexpr int NX = 4, NU = 1, NW = 5, NP = 83, NTD = 3, NTC = 0;
    static constexpr bool DISCRETE = false, HAS_SOFT = false, SKIP_CORRECTOR = true;
    static constexpr bool EXACT_QP = false;   // (true only in CartpoleDevExact below)
#ifndef MPCRL_CARTPOLE_MAX_IPW
#define MPCRL_CARTPOLE_MAX_IPW 4
#endif
    static constexpr int MAX_IPW = MPCRL_CARTPOLE_MAX_IPW;   // instances per wavefront: the matrix-core factor sweep has four 4x4 blocks
#ifndef MPCRL_CARTPOLE_SEG_SKIP
#define MPCRL_CARTPOLE_SEG_SKIP 0
#endif
    static constexpr bool SEG_SKIP = MPCRL_CARTPOLE_SEG_SKIP != 0;   // segmented reductions: full trees (small_kernel.hpp, measured)
    MPCRL_DI static int td_index(int i) { return i; }
    MPCRL_DI static int tc_index(int) { return 0; }
    MPCRL_DI static bool p_has_gradient(int e) { return e < NTD; }   // entries of p the sensitivity kernel computes; the rest are zeros
    // cost block of the parameter vector (nlp.py:969-989, each field column-major): W_0 (5x5), W (5x5), W_e (4x4), yref_0 (5),
    // yref (5), yref_e (4) in y = [x; u] order.  The solve uses whatever set_parameter / cost_set wrote there (mpc.py:233-257);
    // the mirror's cost is not parameterised by them (nlp.py:1039-1055): cost_dp / cost_mixed contribute nothing.
    static constexpr int P_W0 = 3, P_W = 28, P_WE = 53, P_YREF0 = 69, P_YREF = 74, P_YREFE = 79;

    template <class S>
    MPCRL_DI static void ode(const S *x, const S *u, const S *th, S *f) {
        const double g = 9.8;   // config/cartpole.yaml:75-78 (fixed)
        S s, c;
        jsincos(x[2], s, c);
        const S mM = th[1] + th[0];
        const S temp = (u[0] + th[1] * x[3] * x[3] * s) / mM;
        const S thdd = (g * s - c * temp) / (th[2] * (4.0 / 3.0 - th[1] * c * c / mM));
        f[0] = x[1];
        f[1] = temp - th[1] * thdd * c / mM;
        f[2] = x[3];
        f[3] = thdd;
    }
    // Structure of the discrete map: the cart position x0 does not enter the ODE and the cart velocity x1 enters it only through
    // d(x0)/dt = x1, so for RK4 with any step the columns of A for x0 and x1 are e_0 and [T, 1, 0, 0]' (T = step length) exactly.
    // Only u, theta, theta_dot are carried as jet directions.
    __host__ __device__ static constexpr bool soft_coord(int) { return false; }
    static constexpr int NSOFT = 1;                                        // (no soft coordinate: a dummy slot)
    __host__ __device__ static constexpr int soft_slot(int) { return 0; }
    static constexpr int NLD = 3;
    MPCRL_DI static constexpr int lin_coord(int d) { return d == 0 ? 0 : d + 2; }   // stage-vector coordinate (v = [u; x]) of direction d
    template <class F>
    MPCRL_DI static void lin_trivial(double T, F set) {
#pragma unroll
        for (int i = 0; i < NX; ++i) set(i, 0, i == 0 ? 1.0 : 0.0), set(i, 1, i == 0 ? T : (i == 1 ? 1.0 : 0.0));
    }
    MPCRL_DI static constexpr int yi(int i) { return i < NU ? NX + i : i - NU; }   // v index -> y index
    // symmetrised weight between stage-vector coordinates i, j for stage kind 0 (stage 0) / 1 (interior) / 2 (terminal);
    // p = the instance's full parameter vector (global memory; read once per launch into the LDS cost table)
    MPCRL_DI static double hess(int kind, int i, int j, const SmallSpec &, const double *p) {
        if (kind == 2) {
            if (i < NU || j < NU) return 0.0;
            return 0.5 * (p[P_WE + (j - NU) * 4 + (i - NU)] + p[P_WE + (i - NU) * 4 + (j - NU)]);
        }
        const double *W = p + (kind == 0 ? P_W0 : P_W);
        return 0.5 * (W[yi(j) * 5 + yi(i)] + W[yi(i) * 5 + yi(j)]);
    }
    // reference point of the residual, stage-vector order
    MPCRL_DI static double yref(int kind, int i, const double *p) {
        if (kind == 2) return i < NU ? 0.0 : p[P_YREFE + i - NU];
        return p[(kind == 0 ? P_YREF0 : P_YREF) + yi(i)];
    }
    // gradient (stage-vector order) and value of the unscaled stage cost 1/2 r' H r, r = v - yref; H / yr: this stage's set of the
    // cost table (packed lower triangle / reference point)
    MPCRL_DI static double cost_grad(bool, int, const double *x, const double *u, const SmallSpec &, const double *, const double *H,
                                     const double *yr, double *g) {
        double r[NW], val = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) r[i] = (i < NU ? u[i < NU ? i : 0] : x[i >= NU ? i - NU : 0]) - yr[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NW; ++j) a = fma(H[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i], r[j], a);
            g[i] = a;
            val = fma(0.5 * a, r[i], val);
        }
        return val;
    }
    MPCRL_DI static void cost_dp(bool, int, const double *, const double *, double, double *) {}
    MPCRL_DI static void cost_mixed(bool, const double *, double, double *) {}
};

// Test-only model variant behind MPCRL_EXACT_QP (include/mpcrl.h): the cartpole OCP with every QP solved to the tight interior-point
// tolerance — no inexact-SQP forcing term, no interior-point warm start across QPs, fixed fraction to the boundary, no predictor-only
// steps: what acados + HPIPM do with the reference's options (config/cartpole.yaml:8-14), and the device-side twin of the frozen
// exact-QP mode the CPU checker of the test suite has.  A separate instantiation of the plain solve kernel: the shipped kernels compile
// to the code they had without it.
struct CartpoleDevExact : CartpoleDev {
    static constexpr bool SKIP_CORRECTOR = false, EXACT_QP = true;
};

struct LinearDev {
    static constexpr int NX = 2, NU = 1, NW = 3, NP = 12, NTD = 8, NTC = 4;
    static constexpr bool DISCRETE = true, HAS_SOFT = true, SKIP_CORRECTOR = false;
    static constexpr bool EXACT_QP = false;
    static constexpr int MAX_IPW = 21;   // N >= 2
    static constexpr bool SEG_SKIP = true;
    MPCRL_DI static int td_index(int i) { return i; }
    MPCRL_DI static int tc_index(int i) { return 8 + i; }   // V_0, f_0, f_1, f_2
    MPCRL_DI static bool p_has_gradient(int) { return true; }
    __host__ __device__ static constexpr bool soft_coord(int i) { return i == NU; }   // idxsbx = [0]: the first state (linear_system/acados.py:73-131)
    // slack state (s, its multiplier pair, the row slack) is carried for the soft coordinates only: NSOFT slots, soft_slot(i) of coordinate i
    static constexpr int NSOFT = 1;
    __host__ __device__ static constexpr int soft_slot(int) { return 0; }
    static constexpr int NLD = NX + NU;
    MPCRL_DI static constexpr int lin_coord(int d) { return d; }
    template <class F>
    MPCRL_DI static void lin_trivial(double, F) {}
    // consts: P (2x2 row-major)
    template <class S>
    MPCRL_DI static void ode(const S *x, const S *u, const S *th, S *f) {
        // A column-major in p (linear_system/acados.py:60-62,89-90)
        f[0] = th[0] * x[0] + th[2] * x[1] + th[4] * u[0] + th[6];
        f[1] = th[1] * x[0] + th[3] * x[1] + th[5] * u[0] + th[7];
    }
    MPCRL_DI static double hess(int kind, int i, int j, const SmallSpec &sp, const double *) {
        if (kind != 2) return i == j ? 1.0 : 0.0;
        if (i < NU || j < NU) return 0.0;
        return 0.5 * (sp.consts[(i - NU) * 2 + (j - NU)] + sp.consts[(j - NU) * 2 + (i - NU)]);
    }
    MPCRL_DI static double yref(int, int, const double *) { return 0.0; }
    MPCRL_DI static double cost_grad(bool term, int k, const double *x, const double *u, const SmallSpec &sp, const double *tc,
                                     const double *, const double *, double *g) {
        if (!term) {   // l = 1/2 y'y + f'y (+ V_0 at k = 0), y = [x; u], f = tc[1..3]
            g[0] = u[0] + tc[3];
            g[1] = x[0] + tc[1];
            g[2] = x[1] + tc[2];
            const double v = 0.5 * (x[0] * x[0] + x[1] * x[1] + u[0] * u[0]) + tc[1] * x[0] + tc[2] * x[1] + tc[3] * u[0];
            return k == 0 ? v + tc[0] : v;
        }
        const double p00 = sp.consts[0], p01 = 0.5 * (sp.consts[1] + sp.consts[2]), p11 = sp.consts[3];
        g[0] = 0.0;
        g[1] = p00 * x[0] + p01 * x[1];
        g[2] = p01 * x[0] + p11 * x[1];
        return 0.5 * (x[0] * g[1] + x[1] * g[2]);
    }
    // out[NTC] += sc * d l / d (V_0, f)
    MPCRL_DI static void cost_dp(bool term, int k, const double *x, const double *u, double sc, double *out) {
        if (term) return;
        if (k == 0) out[0] += sc;
        out[1] += sc * x[0];
        out[2] += sc * x[1];
        out[3] += sc * u[0];
    }
    // out[NTC] += sc * y' d2 l / dv d(V_0, f),  y in stage-vector order [u; x]
    MPCRL_DI static void cost_mixed(bool term, const double *y, double sc, double *out) {
        if (term) return;
        out[1] += sc * y[1];
        out[2] += sc * y[2];
        out[3] += sc * y[0];
    }
};

// ---------------------------------------------------------------------------------------------------
// Chain of masses (rlmpc/mpc/chain_mass/ocp_utils.py:59-147,253-277), NMASS = n_mass.
// x = [pos (3(M+1)); vel (3M)], u = velocity of the last mass, M = n_mass - 2 free masses.
// p = [m (NL), D (3 NL), L (3 NL), C (3 NL), Q (nx*nx col-major), R (nu*nu col-major), w (3 M)]   (ocp_utils.py:353-371)
// dynamics parameters th = [m, D, L, C, w]; consts = x_ss (NX).
// ---------------------------------------------------------------------------------------------------
template <int NMASS>
struct ChainDev {
    static constexpr int M = NMASS - 2, NL = NMASS - 1;
    static constexpr int NX = (2 * M + 1) * 3, NU = 3, NW = NX + NU;
    static constexpr int OFF_Q = 10 * NL, OFF_R = OFF_Q + NX * NX, OFF_W = OFF_R + NU * NU;
    static constexpr int NP = OFF_W + 3 * M, NTD = 10 * NL + 3 * M;
    MPCRL_DI static int td_index(int i) { return i < 10 * NL ? i : OFF_W + (i - 10 * NL); }

    template <class S>
    MPCRL_DI static void ode(const S *x, const S *u, const S *th, S *f) {
        const S *pos = x, *vel = x + 3 * (M + 1);
        const S *m = th, *D = th + NL, *L = th + 4 * NL, *C = th + 7 * NL, *w = th + 10 * NL;
        S *acc = f + 3 * (M + 1);
        for (int i = 0; i < 3 * M; ++i) acc[i] = (i % 3 == 2) ? w[i] + (-9.81) : w[i];
        for (int i = 0; i <= M; ++i) {
            S dist[3];
            for (int j = 0; j < 3; ++j) dist[j] = i ? pos[3 * i + j] - pos[3 * (i - 1) + j] : pos[j];
            const S nrm = jsqrt(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]);
            for (int j = 0; j < 3; ++j) {
                const S Fs = D[3 * i + j] / m[i] * (1.0 - L[3 * i + j] / nrm) * dist[j];
                const S vr = i < M ? vel[3 * i + j] : u[j];
                const S dv = i ? vr - vel[3 * (i - 1) + j] : vr;
                const S Ft = Fs + C[3 * i + j] * dv;
                if (i < M) acc[3 * i + j] = acc[3 * i + j] - Ft;
                if (i > 0) acc[3 * (i - 1) + j] = acc[3 * (i - 1) + j] + Ft;
            }
        }
        for (int i = 0; i < 3 * M; ++i) f[i] = vel[i];
        for (int j = 0; j < 3; ++j) f[3 * M + j] = u[j];
    }
    // ode() with the parameters as plain doubles read straight from the full parameter vector p (uniform across the lanes of an
    // instance: scalar loads), fully unrolled so that the state arrays of the caller stay in registers.  Used by the linearisation
    // of the solve kernel, where only (x, u) carry tangents.
    template <class S>
    MPCRL_DI static void ode_p(const S *x, const S *u, const double *p, S *f) {
        const S *pos = x, *vel = x + 3 * (M + 1);
        const double *m = p, *D = p + NL, *L = p + 4 * NL, *C = p + 7 * NL, *w = p + OFF_W;
        S *acc = f + 3 * (M + 1);
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) acc[i] = S((i % 3 == 2) ? w[i] + (-9.81) : w[i]);
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            S dist[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dist[j] = i ? pos[3 * i + j] - pos[3 * (i > 0 ? i - 1 : 0) + j] : pos[j];
            const S inrm = jrecip(jsqrt(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]));
            const double im = 1.0 / m[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const S Fs = (D[3 * i + j] * im) * ((1.0 - L[3 * i + j] * inrm) * dist[j]);
                const S vr = i < M ? vel[3 * (i < M ? i : 0) + j] : u[j];
                const S dv = i ? vr - vel[3 * (i > 0 ? i - 1 : 0) + j] : vr;
                const S Ft = Fs + C[3 * i + j] * dv;
                if (i < M) acc[3 * (i < M ? i : 0) + j] = acc[3 * (i < M ? i : 0) + j] - Ft;
                if (i > 0) acc[3 * (i > 0 ? i - 1 : 0) + j] = acc[3 * (i > 0 ? i - 1 : 0) + j] + Ft;
            }
        }
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) f[i] = vel[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) f[3 * M + j] = u[j];
    }
    // Reverse sweep of ode(): xb += (df/dx)' fb, ub += (df/du)' fb and, with WANT_TH, thb += (df/dth)' fb.
    // With S = Jet1<1> (tangent seeded on x, u or fb) this is forward-over-reverse: the tangent parts of xb, ub, thb are one
    // Hessian-vector product of fb' f — what the KKT sensitivities need (nlp.py:1195-1211 builds the same objects symbolically).
    template <bool WANT_TH, class S>
    MPCRL_DI static void ode_adj(const S *x, const S *u, const S *th, const S *fb, S *xb, S *ub, S *thb) {
        const S *pos = x, *vel = x + 3 * (M + 1);
        const S *m = th, *D = th + NL, *L = th + 4 * NL, *C = th + 7 * NL;
        S *posb = xb, *velb = xb + 3 * (M + 1);
        const S *accb = fb + 3 * (M + 1);
        for (int i = 0; i < 3 * M; ++i) velb[i] = velb[i] + fb[i];
        for (int j = 0; j < 3; ++j) ub[j] = ub[j] + fb[3 * M + j];
        if constexpr (WANT_TH)
            for (int i = 0; i < 3 * M; ++i) thb[10 * NL + i] = thb[10 * NL + i] + accb[i];
        const S one(1.0);
        for (int i = 0; i <= M; ++i) {
            S dist[3], distb[3];
            for (int j = 0; j < 3; ++j) dist[j] = i ? pos[3 * i + j] - pos[3 * (i - 1) + j] : pos[j];
            const S inrm = one / jsqrt(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]);
            const S im = one / m[i];
            S nrmb(0.0);
            for (int j = 0; j < 3; ++j) {
                S Ftb(0.0);   // adjoint of the link force: acc_i -= Ft, acc_{i-1} += Ft
                if (i < M) Ftb = Ftb - accb[3 * i + j];
                if (i > 0) Ftb = Ftb + accb[3 * (i - 1) + j];
                const S a = D[3 * i + j] * im;
                const S g = 1.0 - L[3 * i + j] * inrm;
                const S dvb = C[3 * i + j] * Ftb;
                if (i < M) velb[3 * i + j] = velb[3 * i + j] + dvb; else ub[j] = ub[j] + dvb;
                if (i > 0) velb[3 * (i - 1) + j] = velb[3 * (i - 1) + j] - dvb;
                const S gb = Ftb * a * dist[j];
                if constexpr (WANT_TH) {
                    const S vr = i < M ? vel[3 * i + j] : u[j];
                    const S dv = i ? vr - vel[3 * (i - 1) + j] : vr;
                    const S gd = Ftb * g * dist[j] * im;   // dFs/dD
                    thb[7 * NL + 3 * i + j] = thb[7 * NL + 3 * i + j] + Ftb * dv;
                    thb[NL + 3 * i + j] = thb[NL + 3 * i + j] + gd;
                    thb[i] = thb[i] - gd * a;
                    thb[4 * NL + 3 * i + j] = thb[4 * NL + 3 * i + j] - gb * inrm;
                }
                nrmb = nrmb + gb * L[3 * i + j] * inrm * inrm;
                distb[j] = Ftb * a * g;
            }
            for (int j = 0; j < 3; ++j) {
                const S db = distb[j] + nrmb * dist[j] * inrm;
                posb[3 * i + j] = posb[3 * i + j] + db;
                if (i > 0) posb[3 * (i - 1) + j] = posb[3 * (i - 1) + j] - db;
            }
        }
    }
    // ode_adj() with the parameters as plain doubles read from the full parameter vector p (as ode_p): the parameters are never a
    // differentiation direction of the jets (their derivatives come out of the reverse sweep, thb), so carrying them as jets only
    // costs registers — 2 x NTD of them in the forward-over-reverse kernels.  thb is indexed like the dynamics parameters
    // (m, D, L, C, w), i.e. by td position.
    template <bool WANT_TH, class S>
    MPCRL_DI static void ode_adj_p(const S *x, const S *u, const double *p, const S *fb, S *xb, S *ub, S *thb) {
        const S *pos = x, *vel = x + 3 * (M + 1);
        const double *m = p, *D = p + NL, *L = p + 4 * NL, *C = p + 7 * NL;
        S *posb = xb, *velb = xb + 3 * (M + 1);
        const S *accb = fb + 3 * (M + 1);
        for (int i = 0; i < 3 * M; ++i) velb[i] = velb[i] + fb[i];
        for (int j = 0; j < 3; ++j) ub[j] = ub[j] + fb[3 * M + j];
        if constexpr (WANT_TH)
            for (int i = 0; i < 3 * M; ++i) thb[10 * NL + i] = thb[10 * NL + i] + accb[i];
        for (int i = 0; i <= M; ++i) {
            S dist[3], distb[3];
            for (int j = 0; j < 3; ++j) dist[j] = i ? pos[3 * i + j] - pos[3 * (i - 1) + j] : pos[j];
            const S inrm = jrecip(jsqrt(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]));
            const double im = 1.0 / m[i];
            S nrmb(0.0);
            for (int j = 0; j < 3; ++j) {
                S Ftb(0.0);   // adjoint of the link force: acc_i -= Ft, acc_{i-1} += Ft
                if (i < M) Ftb = Ftb - accb[3 * i + j];
                if (i > 0) Ftb = Ftb + accb[3 * (i - 1) + j];
                const double a = D[3 * i + j] * im;
                const S g = 1.0 - L[3 * i + j] * inrm;
                const S dvb = C[3 * i + j] * Ftb;
                if (i < M) velb[3 * i + j] = velb[3 * i + j] + dvb; else ub[j] = ub[j] + dvb;
                if (i > 0) velb[3 * (i - 1) + j] = velb[3 * (i - 1) + j] - dvb;
                const S fd = Ftb * dist[j];
                const S gb = a * fd;
                if constexpr (WANT_TH) {
                    const S vr = i < M ? vel[3 * i + j] : u[j];
                    const S dv = i ? vr - vel[3 * (i - 1) + j] : vr;
                    const S gd = im * (fd * g);   // dFs/dD
                    thb[7 * NL + 3 * i + j] = thb[7 * NL + 3 * i + j] + Ftb * dv;
                    thb[NL + 3 * i + j] = thb[NL + 3 * i + j] + gd;
                    thb[i] = thb[i] - a * gd;
                    thb[4 * NL + 3 * i + j] = thb[4 * NL + 3 * i + j] - gb * inrm;
                }
                nrmb = nrmb + L[3 * i + j] * (gb * inrm * inrm);
                distb[j] = a * (Ftb * g);
            }
            for (int j = 0; j < 3; ++j) {
                const S db = distb[j] + nrmb * dist[j] * inrm;
                posb[3 * i + j] = posb[3 * i + j] + db;
                if (i > 0) posb[3 * (i - 1) + j] = posb[3 * (i - 1) + j] - db;
            }
        }
    }
    // ---- the same ODE split into what depends on the POINT (x, u) and what is linear in a DIRECTION ------------------------------
    // ode_coef: f(x, u) and, per link i, the 9 numbers the tangent of f needs at this point:
    //   dist_i (3),  a_ij = D_ij / m_i (1 - L_ij / n_i),  b_ij = D_ij / m_i  L_ij dist_ij / n_i^3        (n_i = |dist_i|)
    // so that  d Fs_ij = a_ij d dist_ij + b_ij (dist_i . d dist_i).  `tab` ([NL][TAB] doubles, LDS) is written when `store` is set.
    // With SECOND also the second-order pieces  c_ij = D_ij / m_i  L_ij / n_i^3  and  e_i = 3 / n_i^2  (Hessian of the spring force).
    static constexpr int TAB = 9, TAB2 = 13;
    // COMPACT: p holds the NTD differentiable parameters only (td_index order: the disturbance w right after the link coefficients)
    template <bool SECOND, bool COMPACT = false>
    MPCRL_DI static void ode_coef(const double *x, const double *u, const double *p, double *f, double *tab, bool store) {
        constexpr int TS_ = SECOND ? TAB2 : TAB;
        const double *pos = x, *vel = x + 3 * (M + 1);
        const double *m = p, *D = p + NL, *L = p + 4 * NL, *C = p + 7 * NL, *w = p + (COMPACT ? 10 * NL : OFF_W);
        double *acc = f + 3 * (M + 1);
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) acc[i] = (i % 3 == 2) ? w[i] + (-9.81) : w[i];
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            double dist[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dist[j] = i ? pos[3 * i + j] - pos[3 * (i > 0 ? i - 1 : 0) + j] : pos[j];
            const double n2 = dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2];
            const double inrm = 1.0 / sqrt(n2), im = 1.0 / m[i], in3 = inrm * inrm * inrm;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dm = D[3 * i + j] * im, lj = L[3 * i + j];
                const double aj = dm * (1.0 - lj * inrm);
                const double vr = i < M ? vel[3 * (i < M ? i : 0) + j] : u[j];
                const double dv = i ? vr - vel[3 * (i > 0 ? i - 1 : 0) + j] : vr;
                const double Ft = aj * dist[j] + C[3 * i + j] * dv;
                if (i < M) acc[3 * (i < M ? i : 0) + j] -= Ft;
                if (i > 0) acc[3 * (i > 0 ? i - 1 : 0) + j] += Ft;
                if (store) {
                    tab[i * TS_ + j] = dist[j], tab[i * TS_ + 3 + j] = aj, tab[i * TS_ + 6 + j] = dm * lj * dist[j] * in3;
                    if constexpr (SECOND) tab[i * TS_ + 9 + j] = dm * lj * in3;
                }
            }
            if constexpr (SECOND)
                if (store) tab[i * TS_ + 12] = 3.0 * inrm * inrm;
        }
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) f[i] = vel[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) f[3 * M + j] = u[j];
    }
    // ode_tan: df = (df/dx) dx + (df/du) du at the point whose coefficients are in `tab` (stride TS_ per link).  With DD the dist
    // tangents d dist_i of this direction are also returned (the second-order sweep of the sensitivities publishes them).
    template <int TS_, bool DD>
    MPCRL_DI static void ode_tan(const double *tab, const double *p, const double *dx, const double *du, double *df, double *dd) {
        ode_tan_c<TS_, DD>(tab, p + 7 * NL, dx, du, df, dd);
    }
    // (the same with the damping coefficients C_ij — the only parameters the tangent reads — given directly: the direction pass
    // keeps them in LDS next to the tables instead of fetching them from the parameter vector at every evaluation point)
    template <int TS_, bool DD>
    MPCRL_DI static void ode_tan_c(const double *tab, const double *C, const double *dx, const double *du, double *df, double *dd) {
        const double *dpos = dx, *dvel = dx + 3 * (M + 1);
        double *dacc = df + 3 * (M + 1);
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) dacc[i] = 0.0;
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            const double *t = tab + i * TS_;
            double dd_[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dd_[j] = i ? dpos[3 * i + j] - dpos[3 * (i > 0 ? i - 1 : 0) + j] : dpos[j];
            const double dot = t[0] * dd_[0] + t[1] * dd_[1] + t[2] * dd_[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dvr = i < M ? dvel[3 * (i < M ? i : 0) + j] : du[j];
                const double ddv = i ? dvr - dvel[3 * (i > 0 ? i - 1 : 0) + j] : dvr;
                const double dFt = fma(t[3 + j], dd_[j], fma(t[6 + j], dot, C[3 * i + j] * ddv));
                if (i < M) dacc[3 * (i < M ? i : 0) + j] -= dFt;
                if (i > 0) dacc[3 * (i > 0 ? i - 1 : 0) + j] += dFt;
                if constexpr (DD) dd[3 * i + j] = dd_[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) df[i] = dvel[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) df[3 * M + j] = du[j];
    }
    // One link of ode_tan_c with the link's 12 coefficients already in registers (t: dist, a, b of ode_coef, then C_i), accumulating into the
    // acceleration tangents dacc[3 M]: the direction pass fetches the coefficients of the NEXT link from LDS while this one is computed.
    template <int I, bool DD = false>
    MPCRL_DI static void ode_tan_link(const double (&t)[12], const double *dx, const double *du, double *dacc, double *dd = nullptr) {
        const double *dpos = dx, *dvel = dx + 3 * (M + 1);
        double dd_[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dd_[j] = I ? dpos[3 * I + j] - dpos[3 * (I > 0 ? I - 1 : 0) + j] : dpos[j];
        if constexpr (DD) {
#pragma unroll
            for (int j = 0; j < 3; ++j) dd[3 * I + j] = dd_[j];
        }
        const double dot = t[0] * dd_[0] + t[1] * dd_[1] + t[2] * dd_[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double dvr = I < M ? dvel[3 * (I < M ? I : 0) + j] : du[j];
            const double ddv = I ? dvr - dvel[3 * (I > 0 ? I - 1 : 0) + j] : dvr;
            const double dFt = fma(t[3 + j], dd_[j], fma(t[6 + j], dot, t[9 + j] * ddv));
            if (I < M) dacc[3 * (I < M ? I : 0) + j] -= dFt;
            if (I > 0) dacc[3 * (I > 0 ? I - 1 : 0) + j] += dFt;
        }
    }
    // ode_tan_T: the transpose of ode_tan at the same point — xb += (df/dx)' kb (ub is not needed by its caller) — and, per link,
    // the force adjoint q_i = (adjoint of acc_{i-1}) - (adjoint of acc_i), which weights the link's spring force in kb' f.
    template <int TS_>
    MPCRL_DI static void ode_tan_T(const double *tab, const double *p, const double *kb, double *xb, double *q) {
        const double *C = p + 7 * NL;
        double *posb = xb, *velb = xb + 3 * (M + 1);
        const double *accb = kb + 3 * (M + 1);
#pragma unroll
        for (int i = 0; i < 3 * M; ++i) velb[i] += kb[i];
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            const double *t = tab + i * TS_;
            double qi[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double v = 0.0;
                if (i < M) v -= accb[3 * (i < M ? i : 0) + j];
                if (i > 0) v += accb[3 * (i > 0 ? i - 1 : 0) + j];
                qi[j] = v, q[3 * i + j] = v;
            }
            const double bq = t[6] * qi[0] + t[7] * qi[1] + t[8] * qi[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double ddb = fma(t[3 + j], qi[j], t[j] * bq), dvb = C[3 * i + j] * qi[j];
                posb[3 * i + j] += ddb;
                if (i > 0) posb[3 * (i > 0 ? i - 1 : 0) + j] -= ddb;
                if (i < M) velb[3 * (i < M ? i : 0) + j] += dvb;
                if (i > 0) velb[3 * (i > 0 ? i - 1 : 0) + j] -= dvb;
            }
        }
    }
    // Hessian of q' Fs(dist) with respect to dist (3x3 symmetric, packed [00, 10, 11, 20, 21, 22]) from the second-order coefficients
    // of ode_coef<true>: with chat_j = q_j c_j (c_j = D_j / m L_j / n^3) and s = chat . dist,
    //   G = chat dist' + dist chat' + s (I - (3 / n^2) dist dist').
    MPCRL_DI static void link_hessian(const double *t /* TAB2 entries of the link */, const double *q, double *G) {
        const double c0 = q[0] * t[9], c1 = q[1] * t[10], c2 = q[2] * t[11];
        const double s = c0 * t[0] + c1 * t[1] + c2 * t[2], se = s * t[12];
        G[0] = 2.0 * c0 * t[0] + s - se * t[0] * t[0];
        G[1] = c1 * t[0] + t[1] * c0 - se * t[1] * t[0];
        G[2] = 2.0 * c1 * t[1] + s - se * t[1] * t[1];
        G[3] = c2 * t[0] + t[2] * c0 - se * t[2] * t[0];
        G[4] = c2 * t[1] + t[2] * c1 - se * t[2] * t[1];
        G[5] = 2.0 * c2 * t[2] + s - se * t[2] * t[2];
    }
    // symmetrised cost weights from p (Q, R column-major; ocp_utils.py:267,273)
    MPCRL_DI static double Qs(const double *p, int i, int j) { return 0.5 * (p[OFF_Q + j * NX + i] + p[OFF_Q + i * NX + j]); }
    MPCRL_DI static double Rs(const double *p, int i, int j) { return 0.5 * (p[OFF_R + j * NU + i] + p[OFF_R + i * NU + j]); }
    // Hessian of the unscaled stage cost between stage-vector coordinates (v = [u; x])
    MPCRL_DI static double hess(bool term, int i, int j, const double *p) {
        if (i < NU && j < NU) return term ? 0.0 : Rs(p, i, j);
        if (i >= NU && j >= NU) return Qs(p, i - NU, j - NU);
        return 0.0;
    }
};

// RK4^steps with the minimum of live state (used with large NX where the arrays live in scratch)
template <class M, class S>
MPCRL_DI void disc_map_lean(const S *x, const S *u, const S *th, S *xn, double h, int steps) {
    constexpr int NX = M::NX;
    S xc[NX], acc[NX], xt[NX], kk[NX];
    for (int i = 0; i < NX; ++i) xc[i] = x[i];
    for (int s = 0; s < steps; ++s) {
        M::template ode<S>(xc, u, th, kk);
        for (int i = 0; i < NX; ++i) acc[i] = kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode<S>(xt, u, th, kk);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode<S>(xt, u, th, kk);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + h * kk[i];
        M::template ode<S>(xt, u, th, kk);
        for (int i = 0; i < NX; ++i) xc[i] = xc[i] + (h / 6.0) * (acc[i] + kk[i]);
    }
    for (int i = 0; i < NX; ++i) xn[i] = xc[i];
}

// RK4^steps on M::ode_p (parameters = the full vector p as doubles), everything unrolled: no runtime-indexed arrays.
template <class M, class S>
MPCRL_DI void disc_map_p(const S *x, const S *u, const double *p, S *xn, double h, int steps) {
    constexpr int NX = M::NX;
    S xc[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = x[i];
    for (int s = 0; s < steps; ++s) {
        S acc[NX], kk[NX], xt[NX];
        M::template ode_p<S>(xc, u, p, kk);
#pragma unroll
        for (int i = 0; i < NX; ++i) acc[i] = kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode_p<S>(xt, u, p, kk);
#pragma unroll
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode_p<S>(xt, u, p, kk);
#pragma unroll
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + h * kk[i];
        M::template ode_p<S>(xt, u, p, kk);
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = xc[i] + (h / 6.0) * (acc[i] + kk[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = xc[i];
}

// Reverse sweep of F = RK4^steps: xb = (dF/dx)' lam, ub = (dF/du)' lam, thb = (dF/dth)' lam (all overwritten).
// The states at the start of each RK4 step are recomputed rather than stored (steps is 1 or 2 here).
template <class M, bool WANT_TH, class S>
MPCRL_DI void disc_map_adj(const S *x, const S *u, const S *th, const S *lam, S *xb, S *ub, S *thb, double h, int steps) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD;
    S lb[NX];
    for (int i = 0; i < NX; ++i) lb[i] = lam[i];
    for (int i = 0; i < NU; ++i) ub[i] = S(0.0);
    if constexpr (WANT_TH)
        for (int i = 0; i < NTD; ++i) thb[i] = S(0.0);
    for (int s = steps - 1; s >= 0; --s) {
        S xc[NX], X2[NX], X3[NX], X4[NX], kk[NX];
        disc_map_lean<M, S>(x, u, th, xc, h, s);   // state at the start of step s
        M::template ode<S>(xc, u, th, kk);
        for (int i = 0; i < NX; ++i) X2[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode<S>(X2, u, th, kk);
        for (int i = 0; i < NX; ++i) X3[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode<S>(X3, u, th, kk);
        for (int i = 0; i < NX; ++i) X4[i] = xc[i] + h * kk[i];
        S kb[NX], Xb[NX], acc[NX];
        for (int i = 0; i < NX; ++i) kb[i] = (h / 6.0) * lb[i], Xb[i] = S(0.0), acc[i] = lb[i];
        M::template ode_adj<WANT_TH, S>(X4, u, th, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + h * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj<WANT_TH, S>(X3, u, th, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + (0.5 * h) * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj<WANT_TH, S>(X2, u, th, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 6.0) * lb[i] + (0.5 * h) * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj<WANT_TH, S>(xc, u, th, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) lb[i] = acc[i] + Xb[i];
    }
    for (int i = 0; i < NX; ++i) xb[i] = lb[i];
}

// disc_map_adj() on M::ode_p / M::ode_adj_p: parameters as plain doubles out of the full vector p (thb in td order).
template <class M, bool WANT_TH, class S>
MPCRL_DI void disc_map_adj_p(const S *x, const S *u, const double *p, const S *lam, S *xb, S *ub, S *thb, double h, int steps) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD;
    S lb[NX];
    for (int i = 0; i < NX; ++i) lb[i] = lam[i];
    for (int i = 0; i < NU; ++i) ub[i] = S(0.0);
    if constexpr (WANT_TH)
        for (int i = 0; i < NTD; ++i) thb[i] = S(0.0);
    for (int s = steps - 1; s >= 0; --s) {
        S xc[NX], X2[NX], X3[NX], X4[NX], kk[NX];
        disc_map_p<M, S>(x, u, p, xc, h, s);   // state at the start of step s
        M::template ode_p<S>(xc, u, p, kk);
        for (int i = 0; i < NX; ++i) X2[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode_p<S>(X2, u, p, kk);
        for (int i = 0; i < NX; ++i) X3[i] = xc[i] + (0.5 * h) * kk[i];
        M::template ode_p<S>(X3, u, p, kk);
        for (int i = 0; i < NX; ++i) X4[i] = xc[i] + h * kk[i];
        S kb[NX], Xb[NX], acc[NX];
        for (int i = 0; i < NX; ++i) kb[i] = (h / 6.0) * lb[i], Xb[i] = S(0.0), acc[i] = lb[i];
        M::template ode_adj_p<WANT_TH, S>(X4, u, p, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + h * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj_p<WANT_TH, S>(X3, u, p, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + (0.5 * h) * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj_p<WANT_TH, S>(X2, u, p, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 6.0) * lb[i] + (0.5 * h) * Xb[i], Xb[i] = S(0.0);
        M::template ode_adj_p<WANT_TH, S>(xc, u, p, kb, Xb, ub, thb);
        for (int i = 0; i < NX; ++i) lb[i] = acc[i] + Xb[i];
    }
    for (int i = 0; i < NX; ++i) xb[i] = lb[i];
}

// discrete map F = RK4^steps(ode; h)  (rlmpc/common/integrator.py:6-33)
template <class M, class S>
MPCRL_DI void disc_map(const S *x, const S *u, const S *th, S *xn, double h, int steps) {
    constexpr int NX = M::NX;
    if constexpr (M::DISCRETE) {
        M::template ode<S>(x, u, th, xn);
    } else {
        S xc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x[i];
        for (int s = 0; s < steps; ++s) {
            // the weighted sum is accumulated as the stages are produced (same association as k1 + 2 k2 + 2 k3 + k4), so that
            // only one stage derivative is live at a time: 4 NX fewer jets in registers
            S acc[NX], kk[NX], xt[NX];
            M::template ode<S>(xc, u, th, kk);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode<S>(xt, u, th, kk);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode<S>(xt, u, th, kk);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + h * kk[i];
            M::template ode<S>(xt, u, th, kk);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xc[i] + (h / 6.0) * (acc[i] + kk[i]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = xc[i];
    }
}

}  // namespace mpcrl
