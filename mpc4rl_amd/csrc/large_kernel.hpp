// large_kernel.hpp — SQP / Riccati-IPM / adjoint-sensitivity kernels for OCPs whose stage blocks do not fit one lane
// (chain of masses: nx = 21..33, nu = 3, N = 40; rlmpc/mpc/chain_mass/ocp_utils.py).
//
// Mapping on gfx950:
//   * one 256-thread workgroup (4 waves) per OCP instance; every per-stage / per-element phase is thread-strided;
//   * the Riccati recursion keeps the CURRENT stage's P, [B A], P[B A] and the (nx+nu)^2 block in LDS (~21 KB), works on
//     them element-parallel (one output element per thread, inner products of length nx out of LDS) and streams the
//     per-stage results (P_k, K_k, Cholesky factor) to HBM, where the forward sweep and the corrector re-read them:
//     the stage factors of a whole horizon (41 x ~7 KB) do not fit LDS;
//   * derivatives of the 2-step RK4 map are forward-mode jets evaluated one (stage, direction) item per thread.
// Only hard box bounds are supported here (the chain problem has bounds on u only).
//
// The iteration is the one of small_kernel.hpp / DESIGN.md (same constants), so results agree with the oracle to rounding.
#pragma once
#include "small_kernel.hpp"

namespace mpcrl {

constexpr int LARGE_NT = 256;
constexpr int LARGE_MAXNW = 40;

struct LargeSpec {
    int N, np, cost_kind, rk_steps, max_iter;
    double dT, gamma, h, tol;
    double lb0[4], ub0[4];
    double lb[LARGE_MAXNW], ub[LARGE_MAXNW], lbe[LARGE_MAXNW], ube[LARGE_MAXNW];
    const double *consts;   // device: x_ss
};

struct LargeArgs {
    int B, flags, theta_stride;
    const int *perm;
    const double *x0, *u0fix, *theta;
    double *X, *U, *PI, *BND, *RES;   // iterate (layouts of mpcrl_get_iterate)
    double *ws;                       // per-instance workspace, ws_stride doubles each
    size_t ws_stride;
    double *u0_out, *V, *dV, *dpi;
    int *status, *iters;
};

// per-instance workspace layout (doubles)
template <class M>
struct LargeLayout {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD;
    size_t A, B, r, q, dx, du, nuq, Dx, Du, Dnu, rg, rb, rt, Dg, lamw, tw, aff, P, p, K, L, kff, Hex, term, ynu, total;
    __host__ __device__ explicit LargeLayout(int N) {
        size_t o = 0;
        auto take = [&](size_t n) { size_t s = o; o += n; return s; };
        A = take((size_t)N * NX * NX), B = take((size_t)N * NX * NU), r = take((size_t)N * NX), q = take((size_t)(N + 1) * NW);
        dx = take((size_t)(N + 1) * NX), du = take((size_t)N * NU), nuq = take((size_t)(N + 1) * NX);
        Dx = take((size_t)(N + 1) * NX), Du = take((size_t)N * NU), Dnu = take((size_t)(N + 1) * NX);
        rg = take((size_t)(N + 1) * NW), rb = take((size_t)N * NX), rt = take((size_t)(N + 1) * NW), Dg = take((size_t)(N + 1) * NW);
        lamw = take((size_t)2 * (N + 1) * NW), tw = take((size_t)2 * (N + 1) * NW), aff = take((size_t)2 * (N + 1) * NW);
        P = take((size_t)(N + 1) * NX * NX), p = take((size_t)(N + 1) * NX), K = take((size_t)N * NU * NX), L = take((size_t)N * NU * NU);
        kff = take((size_t)N * NU);
        Hex = take((size_t)(N + 1) * NW * NW), term = take((size_t)N * NTD), ynu = take((size_t)(N + 1) * NX);
        total = (o + 7) & ~(size_t)7;
    }
};

// Development aid: -DMPCRL_PROFILE_PHASES accumulates wall-clock ticks (100 MHz) per phase of the interior-point loop
// (read through mpcrl_debug_phases; profiles/microbench/chain_phases.py).  Off in the product build.
#ifdef MPCRL_PROFILE_PHASES
__device__ unsigned long long g_phase_ticks[16];
#define PH_T0() unsigned long long ph_t = wall_clock64()
#define PH(i) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long n_ = wall_clock64(); atomicAdd(&g_phase_ticks[i], n_ - ph_t); ph_t = n_; } } while (0)
#else
#define PH_T0()
#define PH(i)
#endif

MPCRL_DI double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}
MPCRL_DI double wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s));
    return v;
}

template <class M>
struct LargeSolver {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = LARGE_NT;
    const LargeSpec &sp;
    const int N, tid;
    const double *th;   // full parameter vector of this instance
    bool qmode;
    // global (per instance)
    double *X, *U, *NUv;   // NUv: multipliers arriving at stage k, [(N+1)*NX] (index 0 unused) — kept in ws.ynu? no: own array below
    double *A, *Bm, *r, *q, *dx, *du, *nuq, *Dx, *Du, *Dnu, *rg, *rb, *rt, *Dg, *lam, *t, *aff, *P, *p, *K, *L, *kff;
    // LDS
    double *sP, *sBA, *sT, *sM, *sp_, *scc, *smv, *sK, *sL, *sred;

    MPCRL_DI LargeSolver(const LargeSpec &s, int tid_) : sp(s), N(s.N), tid(tid_) {}

    const double *sck;   // LDS: cost scaling c_k, k = 0..N
    MPCRL_DI double ck(int k) const { return sck[k]; }
    MPCRL_DI double ck_eval(int k) const {
        if (sp.cost_kind == 0) return k == N ? 1.0 : sp.dT;                                            // nlp.py:1044-1055
        return k == 0 ? sp.dT : (k == N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);   // nlp.py:1083-1091
    }
    MPCRL_DI double lbv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sp.lb0[i] : -1e30;
        if (k == N) return i >= NU ? sp.lbe[i - NU] : -1e30;
        return sp.lb[i];
    }
    MPCRL_DI double ubv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sp.ub0[i] : 1e30;
        if (k == N) return i >= NU ? sp.ube[i - NU] : 1e30;
        return sp.ub[i];
    }
    MPCRL_DI bool has(int sd, int k, int i) const { return sd ? ubv(k, i) < NO_BOUND : lbv(k, i) > -NO_BOUND; }
    MPCRL_DI bool fixedc(int k, int i) const { return k == 0 && (i >= NU || qmode); }
    MPCRL_DI bool skipc(int k, int i) const { return k == N && i < NU; }
    MPCRL_DI double vc(int k, int i) const { return i < NU ? (k < N ? U[k * NU + i] : 0.0) : X[k * NX + i - NU]; }
    MPCRL_DI double dvc(const double *ax, const double *au, int k, int i) const {
        return i < NU ? (k < N ? au[k * NU + i] : 0.0) : ax[k * NX + i - NU];
    }
    MPCRL_DI double BAg(int k, int m, int j) const { return j < NU ? Bm[(k * NX + m) * NU + j] : A[(k * NX + m) * NX + j - NU]; }

    MPCRL_DI double block_sum(double v) {
        v = wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) sred[tid >> 6] = v;
        __syncthreads();
        return sred[0] + sred[1] + sred[2] + sred[3];
    }
    MPCRL_DI double block_max(double v) {
        v = wave_max(v);
        __syncthreads();
        if ((tid & 63) == 0) sred[tid >> 6] = v;
        __syncthreads();
        return fmax(fmax(sred[0], sred[1]), fmax(sred[2], sred[3]));
    }

    // ---- dynamics linearisation: one (stage, direction) item per thread; returns nothing, fills A, B, r
    MPCRL_DI void linearize_dyn() {
        for (int it = tid; it < N * NW; it += NT) {
            const int k = it / NW, d = it - k * NW;
            Jet1<1> jx[NX], ju[NU], jt[NTD], jn[NX];
            for (int i = 0; i < NU; ++i) ju[i] = Jet1<1>(U[k * NU + i]);
            for (int i = 0; i < NX; ++i) jx[i] = Jet1<1>(X[k * NX + i]);
            for (int i = 0; i < NTD; ++i) jt[i] = Jet1<1>(th[M::td_index(i)]);
            if (d < NU)
                ju[d].d[0] = 1.0;
            else
                jx[d - NU].d[0] = 1.0;
            disc_map_lean<M, Jet1<1>>(jx, ju, jt, jn, sp.h, sp.rk_steps);
            for (int i = 0; i < NX; ++i) {
                if (d < NU)
                    Bm[(k * NX + i) * NU + d] = jn[i].d[0];
                else
                    A[(k * NX + i) * NX + d - NU] = jn[i].d[0];
                if (d == 0) r[k * NX + i] = jn[i].v - X[(k + 1) * NX + i];
            }
        }
    }
    // cost gradient q = c_k grad l_k and local cost value
    MPCRL_DI double linearize_cost() {
        const double *xs = sp.consts;
        double val = 0.0;
        for (int e = tid; e < (N + 1) * NW; e += NT) {
            const int k = e / NW, i = e - k * NW;
            const bool term = k == N;
            double a = 0.0;
            if (i < NU) {
                if (!term)
                    for (int j = 0; j < NU; ++j) a = fma(M::Rs(th, i, j), U[k * NU + j], a);
                q[e] = ck(k) * a;
                if (!term) val += 0.5 * ck(k) * a * U[k * NU + i];
            } else {
                for (int j = 0; j < NX; ++j) a = fma(M::Qs(th, i - NU, j), X[k * NX + j] - xs[j], a);
                q[e] = ck(k) * a;
                val += 0.5 * ck(k) * a * (X[k * NX + i - NU] - xs[i - NU]);
            }
        }
        return val;
    }
    MPCRL_DI double GTnu(const double *nu, int k, int i) const {
        double a = 0.0;
        if (k < N)
            for (int m = 0; m < NX; ++m) a = fma(BAg(k, m, i), nu[(k + 1) * NX + m], a);
        if (i >= NU && k > 0) a -= nu[k * NX + i - NU];
        return a;
    }
    MPCRL_DI void nlp_residuals(const double *x0, const double *u0f, double *res) {
        double rs = 0, re = 0, ri = 0, rc = 0;
        for (int e = tid; e < (N + 1) * NW; e += NT) {
            const int k = e / NW, i = e - k * NW;
            if (skipc(k, i)) continue;
            if (!fixedc(k, i)) {
                double g = q[e] + GTnu(NUv, k, i);
                if (has(0, k, i)) g -= lam[e];
                if (has(1, k, i)) g += lam[(N + 1) * NW + e];
                rs = fmax(rs, fabs(g));
            }
            const double v = vc(k, i);
            if (has(0, k, i)) {
                const double h = lbv(k, i) - v;
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[e] * h));
            }
            if (has(1, k, i)) {
                const double h = v - ubv(k, i);
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[(N + 1) * NW + e] * h));
            }
        }
        for (int e = tid; e < N * NX; e += NT) re = fmax(re, fabs(r[e]));
        if (tid < NX) re = fmax(re, fabs(X[tid] - x0[tid]));
        if (qmode && tid < NU) re = fmax(re, fabs(U[tid] - u0f[tid]));
        res[0] = block_max(rs), res[1] = block_max(re), res[2] = block_max(ri), res[3] = block_max(rc);
    }

    // ---- backward Riccati sweep.  FACTOR: matrix + vector recursion; else vector only (re-reads P, K, L from HBM).
    // Hs(k,i,j): scaled stage Hessian accessor; g: modified gradient [(N+1)*NW]; bb: dynamics offsets [N*NX] or null (= 0)
    template <bool FACTOR, class HF>
    MPCRL_DI bool backward(HF Hs, const double *g, const double *bb) {
        bool ok = true;
        // terminal stage
        if constexpr (FACTOR) {
            for (int e = tid; e < NX * NX; e += NT) {
                const int i = e / NX, j = e - i * NX;
                const double v = Hs(N, NU + i, NU + j) + (i == j ? Dg[N * NW + NU + i] : 0.0);
                sP[e] = v;
                P[(size_t)N * NX * NX + e] = v;
            }
        }
        if (tid < NX) {
            sp_[tid] = g[N * NW + NU + tid];
            p[N * NX + tid] = sp_[tid];
        }
        __syncthreads();
        for (int k = N - 1; k >= 0; --k) {
            const bool pin = k == 0 && qmode;
            if constexpr (FACTOR) {
                for (int e = tid; e < NX * NW; e += NT) {
                    const int m = e / NW, j = e - m * NW;
                    sBA[e] = BAg(k, m, j);
                }
                __syncthreads();
                // T = P [B A]; cc = p + P bb
                for (int e = tid; e < NX * NW + NX; e += NT) {
                    if (e < NX * NW) {
                        const int i = e / NW, j = e - i * NW;
                        double a = 0.0;
                        for (int m = 0; m < NX; ++m) a = fma(sP[i * NX + m], sBA[m * NW + j], a);
                        sT[e] = a;
                    } else {
                        const int i = e - NX * NW;
                        double a = sp_[i];
                        if (bb)
                            for (int m = 0; m < NX; ++m) a = fma(sP[i * NX + m], bb[k * NX + m], a);
                        scc[i] = a;
                    }
                }
                __syncthreads();
                // M = H + D + [B A]' T (lower triangle, mirrored); mv = g + [B A]' cc
                for (int e = tid; e < NW * NW + NW; e += NT) {
                    if (e < NW * NW) {
                        const int i = e / NW, j = e - i * NW;
                        if (j <= i) {
                            double a = Hs(k, i, j) + (i == j ? Dg[k * NW + i] : 0.0);
                            for (int m = 0; m < NX; ++m) a = fma(sBA[m * NW + i], sT[m * NW + j], a);
                            sM[i * NW + j] = a;
                            sM[j * NW + i] = a;
                        }
                    } else {
                        const int i = e - NW * NW;
                        double a = g[k * NW + i];
                        for (int m = 0; m < NX; ++m) a = fma(sBA[m * NW + i], scc[m], a);
                        smv[i] = a;
                    }
                }
                __syncthreads();
                // Cholesky of the control block (one thread), L lower with inverted diagonal
                if (tid == 0) {
                    bool okc = true;
                    for (int i = 0; i < NU; ++i)
                        for (int j = 0; j <= i; ++j) {
                            double a = sM[i * NW + j];
                            for (int m = 0; m < j; ++m) a -= sL[i * NU + m] * sL[j * NU + m];
                            if (i == j) {
                                okc = okc && (a > 0.0);
                                sL[i * NU + i] = 1.0 / sqrt(a);
                            } else
                                sL[i * NU + j] = a * sL[j * NU + j];
                        }
                    sred[8] = (okc || pin) ? 0.0 : 1.0;
                }
                __syncthreads();
                ok = ok && sred[8] == 0.0;
                // K columns (and the feed-forward as column NX): solve L L' y = rhs
                if (tid <= NX) {
                    const int j = tid;
                    double y[NU], z[NU];
                    for (int i = 0; i < NU; ++i) {
                        double a = j < NX ? sM[(NU + j) * NW + i] : smv[i];
                        for (int m = 0; m < i; ++m) a -= sL[i * NU + m] * y[m];
                        y[i] = a * sL[i * NU + i];
                    }
                    for (int i = NU - 1; i >= 0; --i) {
                        double a = y[i];
                        for (int m = i + 1; m < NU; ++m) a -= sL[m * NU + i] * z[m];
                        z[i] = a * sL[i * NU + i];
                    }
                    for (int i = 0; i < NU; ++i) {
                        const double v = pin ? 0.0 : z[i];
                        if (j < NX) {
                            sK[i * NX + j] = v;
                            K[(k * NU + i) * NX + j] = v;
                        } else {
                            sK[NU * NX + i] = v;
                            kff[k * NU + i] = v;
                        }
                    }
                }
                if (tid < NU * NU) L[k * NU * NU + tid] = sL[tid];
                __syncthreads();
                // P_k = Q - S' K (lower, mirrored); p_k = mv_x - K' mv_u
                for (int e = tid; e < NX * NX + NX; e += NT) {
                    if (e < NX * NX) {
                        const int i = e / NX, j = e - i * NX;
                        if (j <= i) {
                            double a = sM[(NU + i) * NW + NU + j];
                            for (int m = 0; m < NU; ++m) a -= sM[(NU + i) * NW + m] * sK[m * NX + j];
                            sP[i * NX + j] = a, sP[j * NX + i] = a;
                            P[(size_t)k * NX * NX + i * NX + j] = a, P[(size_t)k * NX * NX + j * NX + i] = a;
                        }
                    } else {
                        const int i = e - NX * NX;
                        double a = smv[NU + i];
                        for (int m = 0; m < NU; ++m) a -= sK[m * NX + i] * smv[m];
                        sp_[i] = a;
                        p[k * NX + i] = a;
                    }
                }
                __syncthreads();
            } else {
                // vector recursion only: cc = p_{k+1} + P_{k+1} bb
                if (tid < NX) {
                    double a = sp_[tid];
                    if (bb) {
                        const double *Pn = P + (size_t)(k + 1) * NX * NX;
                        for (int m = 0; m < NX; ++m) a = fma(Pn[tid * NX + m], bb[k * NX + m], a);
                    }
                    scc[tid] = a;
                }
                __syncthreads();
                if (tid < NW) {
                    double a = g[k * NW + tid];
                    for (int m = 0; m < NX; ++m) a = fma(BAg(k, m, tid), scc[m], a);
                    smv[tid] = a;
                }
                __syncthreads();
                if (tid == 0) {
                    const double *Lk = L + k * NU * NU;
                    double y[NU], z[NU];
                    for (int i = 0; i < NU; ++i) {
                        double a = smv[i];
                        for (int m = 0; m < i; ++m) a -= Lk[i * NU + m] * y[m];
                        y[i] = a * Lk[i * NU + i];
                    }
                    for (int i = NU - 1; i >= 0; --i) {
                        double a = y[i];
                        for (int m = i + 1; m < NU; ++m) a -= Lk[m * NU + i] * z[m];
                        z[i] = a * Lk[i * NU + i];
                    }
                    for (int i = 0; i < NU; ++i) {
                        const double v = pin ? 0.0 : z[i];
                        sK[NU * NX + i] = v;
                        kff[k * NU + i] = v;
                    }
                }
                __syncthreads();
                if (tid < NX) {
                    double a = smv[NU + tid];
                    for (int m = 0; m < NU; ++m) a -= K[(k * NU + m) * NX + tid] * smv[m];
                    sp_[tid] = a;
                    p[k * NX + tid] = a;
                }
                __syncthreads();
            }
        }
        return ok;
    }

    // ---- forward sweep: Dx, Du (serial over stages), then Dnu for all stages in parallel
    MPCRL_DI void forward(const double *bb, bool want_nu = true) {
        if (tid < NX) scc[tid] = 0.0, Dx[tid] = 0.0;
        __syncthreads();
        for (int k = 0; k < N; ++k) {
            if (tid < NU) {
                double a = -kff[k * NU + tid];
                for (int j = 0; j < NX; ++j) a = fma(-K[(k * NU + tid) * NX + j], scc[j], a);
                smv[tid] = a;
                Du[k * NU + tid] = a;
            }
            __syncthreads();
            double xn = 0.0;
            if (tid < NX) {
                xn = bb ? bb[k * NX + tid] : 0.0;
                for (int j = 0; j < NX; ++j) xn = fma(A[(k * NX + tid) * NX + j], scc[j], xn);
                for (int j = 0; j < NU; ++j) xn = fma(Bm[(k * NX + tid) * NU + j], smv[j], xn);
            }
            __syncthreads();
            if (tid < NX) scc[tid] = xn, Dx[(k + 1) * NX + tid] = xn;
            __syncthreads();
        }
        for (int e = tid; want_nu && e < (N + 1) * NX; e += NT) {
            const int k = e / NX, i = e - k * NX;
            double a = 0.0;
            if (k > 0) {
                a = p[e];
                const double *Pk = P + (size_t)k * NX * NX;
                for (int j = 0; j < NX; ++j) a = fma(Pk[i * NX + j], Dx[k * NX + j], a);
            }
            Dnu[e] = a;
        }
        __syncthreads();
    }

    // ---- interior point rows (hard bounds) --------------------------------------------------------
    MPCRL_DI double &LAM(int sd, int e) { return lam[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &TT(int sd, int e) { return t[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &AFF(int sd, int e) { return aff[sd * (N + 1) * NW + e]; }
    MPCRL_DI double bslack(int sd, int k, int i, double v) const { return sd ? ubv(k, i) - v : v - lbv(k, i); }

    MPCRL_DI bool qp_solve(const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu) {
        const bool warm = warm_mu > 0.0;
        auto Hs = [&](int k, int i, int j) { return ck(k) * M::hess(k == N, i, j, th); };
        const int ne = (N + 1) * NW;
        for (int e = tid; e < (N + 1) * NX; e += NT) dx[e] = e < NX ? x0[e] - X[e] : 0.0, nuq[e] = warm ? NUv[e] : 0.0;
        for (int e = tid; e < N * NU; e += NT) du[e] = (qmode && e < NU) ? u0f[e] - U[e] : 0.0;
        __syncthreads();
        double cnt = 0.0;
        for (int e = tid; e < ne; e += NT) {
            const int k = e / NW, i = e - k * NW;
            if (skipc(k, i)) continue;
            const double v = vc(k, i) + dvc(dx, du, k, i);
            for (int sd = 0; sd < 2; ++sd)
                if (has(sd, k, i)) {
                    cnt += 1.0;
                    if (warm) {
                        double l = LAM(sd, e), tt = fmax(bslack(sd, k, i, v), TT(sd, e));
                        if (l * tt < warm_mu) {
                            if (l >= tt)
                                tt = warm_mu / l;
                            else
                                l = warm_mu / tt;
                        }
                        LAM(sd, e) = l, TT(sd, e) = tt;
                    } else {
                        TT(sd, e) = fmax(bslack(sd, k, i, v), IPM_T_MIN);
                        LAM(sd, e) = IPM_MU0 / TT(sd, e);
                    }
                }
        }
        const double n_rows = block_sum(cnt);
        bool ok = false;
        PH_T0();
        for (int it = 0;; ++it) {
            PH(7);
            double rloc = 0.0, muloc = 0.0;
            for (int e = tid; e < N * NX; e += NT) {
                const int k = e / NX, i = e - k * NX;
                double a = r[e] - dx[(k + 1) * NX + i];
                for (int j = 0; j < NX; ++j) a = fma(A[(k * NX + i) * NX + j], dx[k * NX + j], a);
                for (int j = 0; j < NU; ++j) a = fma(Bm[(k * NX + i) * NU + j], du[k * NU + j], a);
                rb[e] = a;
                rloc = fmax(rloc, fabs(a));
            }
            for (int e = tid; e < ne; e += NT) {
                const int k = e / NW, i = e - k * NW;
                double a = 0.0;
                if (!skipc(k, i)) {
                    a = q[e] + GTnu(nuq, k, i);
                    for (int j = 0; j < NW; ++j) {
                        const double hij = Hs(k, i, j);
                        if (hij != 0.0) a = fma(hij, dvc(dx, du, k, j), a);
                    }
                    if (has(0, k, i)) a -= LAM(0, e);
                    if (has(1, k, i)) a += LAM(1, e);
                    if (fixedc(k, i)) a = 0.0;
                    const double v = vc(k, i) + dvc(dx, du, k, i);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has(sd, k, i)) {
                            rloc = fmax(rloc, fabs(TT(sd, e) - bslack(sd, k, i, v)));
                            muloc = fma(LAM(sd, e), TT(sd, e), muloc);
                        }
                }
                rg[e] = a;
                rloc = fmax(rloc, fabs(a));
            }
            const double rinf = block_max(rloc);
            const double mu = n_rows > 0.0 ? block_sum(muloc) / n_rows : 0.0;
            if (rinf <= tol_res && mu <= tol_mu) {
                ok = true;
                break;
            }
            if (it >= IPM_MAX_ITER || !(rinf < 1e300)) break;
            ++n_it;
            PH(0);
            double sigma_mu = 0.0, alpha = 1.0;
            bool fail = false;
            for (int pass = 0; pass < 2; ++pass) {
                // barrier diagonal + modified gradient
                for (int e = tid; e < ne; e += NT) {
                    const int k = e / NW, i = e - k * NW;
                    double dg = 0.0, er = 0.0;
                    if (!skipc(k, i)) {
                        const double v = vc(k, i) + dvc(dx, du, k, i);
                        for (int sd = 0; sd < 2; ++sd)
                            if (has(sd, k, i)) {
                                const double l1 = LAM(sd, e), t1 = TT(sd, e);
                                const double rd1 = t1 - bslack(sd, k, i, v);
                                const double rm = fma(l1, t1, pass ? AFF(sd, e) - sigma_mu : 0.0);
                                dg += l1 / t1;
                                er += (sd ? -1.0 : 1.0) * (rm - l1 * rd1) / t1;
                            }
                    }
                    if (pass == 0) Dg[e] = dg;
                    rt[e] = rg[e] + er;
                }
                __syncthreads();
                PH(1);
                if (pass == 0) {
                    if (!backward<true>(Hs, rt, rb)) fail = true;
                    PH(2);
                } else {
                    backward<false>(Hs, rt, rb);
                    PH(3);
                }
                forward(rb, pass == 1);   // the multiplier step is only needed with the final direction
                PH(4);
                double amax = 1.0;
                for (int e = tid; e < ne; e += NT) {
                    const int k = e / NW, i = e - k * NW;
                    if (skipc(k, i)) continue;
                    const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has(sd, k, i)) {
                            const double l1 = LAM(sd, e), t1 = TT(sd, e);
                            const double rd1 = t1 - bslack(sd, k, i, v);
                            const double rm = fma(l1, t1, pass ? AFF(sd, e) - sigma_mu : 0.0);
                            const double dt1 = -rd1 + (sd ? -dv : dv);
                            const double dl1 = (-rm - l1 * dt1) / t1;
                            if (dl1 < 0.0) amax = fmin(amax, -l1 / dl1);
                            if (dt1 < 0.0) amax = fmin(amax, -t1 / dt1);
                        }
                }
                amax = -block_max(-amax);
                if (pass == 0) {
                    double muaff = 0.0;
                    for (int e = tid; e < ne; e += NT) {
                        const int k = e / NW, i = e - k * NW;
                        if (skipc(k, i)) continue;
                        const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                        for (int sd = 0; sd < 2; ++sd)
                            if (has(sd, k, i)) {
                                const double l1 = LAM(sd, e), t1 = TT(sd, e);
                                const double rd1 = t1 - bslack(sd, k, i, v);
                                const double dt1 = -rd1 + (sd ? -dv : dv);
                                const double dl1 = (-l1 * t1 - l1 * dt1) / t1;
                                muaff = fma(fma(amax, dl1, l1), fma(amax, dt1, t1), muaff);
                                AFF(sd, e) = dl1 * dt1;
                            }
                    }
                    const double mu_aff = n_rows > 0.0 ? block_sum(muaff) / n_rows : 0.0;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                } else
                    alpha = fmin(1.0, fmax(IPM_FRAC, 1.0 - mu) * amax);   // fraction to the boundary -> 1 as mu -> 0
            }
            PH(5);
            if (fail) break;
            for (int e = tid; e < ne; e += NT) {
                const int k = e / NW, i = e - k * NW;
                if (skipc(k, i)) continue;
                const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                for (int sd = 0; sd < 2; ++sd)
                    if (has(sd, k, i)) {
                        const double l1 = LAM(sd, e), t1 = TT(sd, e);
                        const double rd1 = t1 - bslack(sd, k, i, v);
                        const double rm = fma(l1, t1, AFF(sd, e) - sigma_mu);
                        const double dt1 = -rd1 + (sd ? -dv : dv);
                        const double dl1 = (-rm - l1 * dt1) / t1;
                        LAM(sd, e) = fma(alpha, dl1, l1);
                        TT(sd, e) = fma(alpha, dt1, t1);
                    }
            }
            __syncthreads();
            for (int e = tid; e < (N + 1) * NX; e += NT) dx[e] = fma(alpha, Dx[e], dx[e]), nuq[e] = fma(alpha, Dnu[e], nuq[e]);
            for (int e = tid; e < N * NU; e += NT) du[e] = fma(alpha, Du[e], du[e]);
            __syncthreads();
        }
        return ok;
    }
};

// =====================================================================================================
// solve kernel: one workgroup per instance
// =====================================================================================================
template <class M>
__global__ void __launch_bounds__(LARGE_NT, 4) large_solve_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU;
    __shared__ double sP[NX * NX], sBA[NX * NW], sT[NX * NW], sM[NW * NW];
    __shared__ double sp_[NX], scc[NX], smv[NW], sK[NU * NX + NU], sL[NU * NU], sred[16], sck[64];
    const int tid = threadIdx.x, inst = a.perm ? a.perm[blockIdx.x] : blockIdx.x, N = sp.N;
    LargeSolver<M> S(sp, tid);
    S.sP = sP, S.sBA = sBA, S.sT = sT, S.sM = sM, S.sp_ = sp_, S.scc = scc, S.smv = smv, S.sK = sK, S.sL = sL, S.sred = sred;
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    S.A = w + lay.A, S.Bm = w + lay.B, S.r = w + lay.r, S.q = w + lay.q, S.dx = w + lay.dx, S.du = w + lay.du, S.nuq = w + lay.nuq;
    S.Dx = w + lay.Dx, S.Du = w + lay.Du, S.Dnu = w + lay.Dnu, S.rg = w + lay.rg, S.rb = w + lay.rb, S.rt = w + lay.rt, S.Dg = w + lay.Dg;
    S.lam = w + lay.lamw, S.t = w + lay.tw, S.aff = w + lay.aff, S.P = w + lay.P, S.p = w + lay.p, S.K = w + lay.K, S.L = w + lay.L;
    S.kff = w + lay.kff, S.NUv = w + lay.ynu, S.sck = sck;
    if (tid <= N) sck[tid] = S.ck_eval(tid);
    __syncthreads();
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    double *bnd = a.BND + (size_t)inst * 10 * nb;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = S.qmode ? a.u0fix + (size_t)inst * NU : nullptr;
    const int ne = (N + 1) * NW;
    // ---- iterate: cold start (MPC.reset, mpc.py:204-210) or the stored one
    if (a.flags & 8) {
        for (int e = tid; e < (N + 1) * NX; e += LARGE_NT) S.X[e] = x0[e % NX], S.NUv[e] = 0.0;
        for (int e = tid; e < N * NU; e += LARGE_NT) S.U[e] = 0.0;
        for (int e = tid; e < 2 * ne; e += LARGE_NT) S.lam[e] = 0.0, S.t[e] = 1.0, S.aff[e] = 0.0;
    } else {
        for (int e = tid; e < (N + 1) * NX; e += LARGE_NT) S.NUv[e] = e < NX ? 0.0 : PIg[e - NX];
        for (int e = tid; e < 2 * ne; e += LARGE_NT) S.lam[e] = bnd[e], S.t[e] = bnd[2 * nb + e], S.aff[e] = 0.0;
    }
    __syncthreads();
    const bool rti = (a.flags & 4) != 0;
    const int max_iter = rti ? 1 : sp.max_iter;
    int status = 2, n_sqp = 0, n_ipm = 0;
    double cost = 0.0, res[4] = {0, 0, 0, 0};
    bool last_tight = true;
    double stepn = -1.0;   // perturbation seen by the next QP (< 0: cold)
    if (!(a.flags & 8)) {
        double sl = 0.0;
        if (tid < NX) sl = fabs(x0[tid] - S.X[tid]);
        if (S.qmode && tid < NU) sl = fmax(sl, fabs(u0f[tid] - S.U[tid]));
        stepn = S.block_max(sl);
    }
    for (int it = 0;; ++it) {
        S.linearize_dyn();
        const double cl = S.linearize_cost();
        __syncthreads();
        cost = S.block_sum(cl);
        S.nlp_residuals(x0, u0f, res);
        n_sqp = it;
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        if (!(rmax < 1e300)) {
            status = 1;
            break;
        }
        if (rmax < sp.tol && last_tight && !(rti && it == 0)) {
            status = 0;
            break;
        }
        if (it >= max_iter) {
            status = rmax < sp.tol ? 0 : 2;
            break;
        }
        const double rr_ = fmin(1.0, rmax), ad_ = rmax < sp.tol ? 0.0 : IPM_ADAPT_C * rr_ * rr_;
        const double tol_res = fmin(IPM_ADAPT_CAP, fmax(IPM_TOL_RES, ad_)), tol_mu = fmin(0.1 * IPM_ADAPT_CAP, fmax(IPM_TOL_MU, 1e-2 * ad_));
        last_tight = tol_res <= IPM_TOL_RES && tol_mu <= IPM_TOL_MU;
        const double warm_mu = stepn < 0.0 ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
        if (!S.qp_solve(x0, u0f, n_ipm, warm_mu, tol_res, tol_mu)) {
            status = 4;
            break;
        }
        {
            double sl = 0.0;
            for (int e = tid; e < (N + 1) * NX; e += LARGE_NT) sl = fmax(sl, fabs(S.dx[e]));
            for (int e = tid; e < N * NU; e += LARGE_NT) sl = fmax(sl, fabs(S.du[e]));
            stepn = S.block_max(sl);
        }
        for (int e = tid; e < (N + 1) * NX; e += LARGE_NT) S.X[e] += S.dx[e], S.NUv[e] = S.nuq[e];
        for (int e = tid; e < N * NU; e += LARGE_NT) S.U[e] += S.du[e];
        __syncthreads();
    }
    // ---- results + iterate
    if (tid < NU) a.u0_out[(size_t)inst * NU + tid] = S.U[tid];
    if (tid == 0) {
        a.V[inst] = cost;
        a.status[inst] = status;
        if (a.iters) a.iters[inst * 2] = n_sqp, a.iters[inst * 2 + 1] = n_ipm;
        for (int j = 0; j < 4; ++j) a.RES[(size_t)inst * 4 + j] = res[j];
    }
    for (int e = tid; e < N * NX; e += LARGE_NT) PIg[e] = S.NUv[NX + e];
    for (int e = tid; e < 2 * ne; e += LARGE_NT) {
        const int sd = e / ne, ee = e - sd * ne, k = ee / NW, i = ee - k * NW;
        const bool h = !S.skipc(k, i) && S.has(sd, k, i);
        bnd[e] = h ? S.lam[e] : 0.0;
        bnd[2 * nb + e] = h ? S.t[e] : 1.0;
    }
    for (int e = tid; e < 6 * ne; e += LARGE_NT) bnd[4 * nb + e] = (e >= 4 * ne) ? 1.0 : 0.0;   // no soft rows here
}

// =====================================================================================================
// sensitivity kernel (dV/dp = dL/dp, nlp.py:1211,1401; du0*/dp by an adjoint Riccati solve, nlp.py:1413-1424).
// Re-uses A, B, q, lam, t, nu left in the workspace by large_solve_kernel (its last linearisation is at the final iterate).
// =====================================================================================================
template <class M>
__global__ void __launch_bounds__(LARGE_NT) large_sens_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = LARGE_NT;
    __shared__ double sP[NX * NX], sBA[NX * NW], sT[NX * NW], sM[NW * NW];
    __shared__ double sp_[NX], scc[NX], smv[NW], sK[NU * NX + NU], sL[NU * NU], sred[16], sck[64];
    const int tid = threadIdx.x, inst = a.perm ? a.perm[blockIdx.x] : blockIdx.x, N = sp.N;
    LargeSolver<M> S(sp, tid);
    S.sP = sP, S.sBA = sBA, S.sT = sT, S.sM = sM, S.sp_ = sp_, S.scc = scc, S.smv = smv, S.sK = sK, S.sL = sL, S.sred = sred;
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    S.A = w + lay.A, S.Bm = w + lay.B, S.r = w + lay.r, S.q = w + lay.q, S.dx = w + lay.dx, S.du = w + lay.du, S.nuq = w + lay.nuq;
    S.Dx = w + lay.Dx, S.Du = w + lay.Du, S.Dnu = w + lay.Dnu, S.rg = w + lay.rg, S.rb = w + lay.rb, S.rt = w + lay.rt, S.Dg = w + lay.Dg;
    S.lam = w + lay.lamw, S.t = w + lay.tw, S.aff = w + lay.aff, S.P = w + lay.P, S.p = w + lay.p, S.K = w + lay.K, S.L = w + lay.L;
    S.kff = w + lay.kff, S.NUv = w + lay.ynu, S.sck = sck;
    if (tid <= N) sck[tid] = S.ck_eval(tid);
    __syncthreads();
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    double *bnd = a.BND + (size_t)inst * 10 * nb;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = S.qmode ? a.u0fix + (size_t)inst * NU : nullptr;
    const int ne = (N + 1) * NW;
    (void)PIg, (void)bnd, (void)x0, (void)u0f;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const double *th = S.th, *xs = sp.consts, *nu = S.NUv;
    double *Hex = w + lay.Hex, *term = w + lay.term;
    // ---- dV/dtheta of the dynamics: sum_k grad_theta (nu_{k+1}' F_k) by one reverse sweep of F per stage
    for (int k = tid; k < N; k += NT) {
        double jx[NX], ju[NU], jt[NTD], lm[NX], xb[NX], ub[NU], tb[NTD];
        for (int i = 0; i < NU; ++i) ju[i] = S.U[k * NU + i];
        for (int i = 0; i < NX; ++i) jx[i] = S.X[k * NX + i], lm[i] = nu[(k + 1) * NX + i];
        for (int i = 0; i < NTD; ++i) jt[i] = th[M::td_index(i)];
        disc_map_adj<M, true, double>(jx, ju, jt, lm, xb, ub, tb, sp.h, sp.rk_steps);
        for (int d = 0; d < NTD; ++d) term[k * NTD + d] = tb[d];
    }
    __syncthreads();
    if ((a.flags & 1) && a.dV) {
        double *dV = a.dV + (size_t)inst * NP;
        for (int d = tid; d < NTD; d += NT) {
            double acc = 0.0;
            for (int k = 0; k < N; ++k) acc += term[k * NTD + d];
            dV[M::td_index(d)] = acc;
        }
        for (int e = tid; e < NX * NX + NU * NU; e += NT) {   // d/dQ_ij, d/dR_ij of sum_k c_k l_k (ocp_utils.py:276-277)
            double acc = 0.0;
            if (e < NX * NX) {
                const int j = e / NX, i = e - j * NX;          // column-major position of Q(i, j)
                for (int k = 0; k <= N; ++k) acc = fma(0.5 * S.ck(k) * (S.X[k * NX + i] - xs[i]), S.X[k * NX + j] - xs[j], acc);
                dV[M::OFF_Q + e] = acc;
            } else {
                const int ee = e - NX * NX, j = ee / NU, i = ee - j * NU;
                for (int k = 0; k < N; ++k) acc = fma(0.5 * S.ck(k) * S.U[k * NU + i], S.U[k * NU + j], acc);
                dV[M::OFF_R + ee] = acc;
            }
        }
    }
    if (!((a.flags & 2) && a.dpi) || S.qmode) return;
    double *dpi = a.dpi + (size_t)inst * NU * NP;
    // ---- exact Lagrangian Hessian blocks c_k hess l + hess (nu_{k+1}' F_k): one (stage, column) item per thread, the column by
    // forward-over-reverse (tangent e_j through the reverse sweep of F)
    __syncthreads();   // term is re-used below
    for (int it = tid; it < N * NW; it += NT) {
        const int k = it / NW, j = it - k * NW;
        Jet1<1> jx[NX], ju[NU], jt[NTD], lm[NX], xb[NX], ub[NU];
        for (int c = 0; c < NU; ++c) ju[c] = Jet1<1>(S.U[k * NU + c]);
        for (int c = 0; c < NX; ++c) jx[c] = Jet1<1>(S.X[k * NX + c]), lm[c] = Jet1<1>(nu[(k + 1) * NX + c]);
        for (int c = 0; c < NTD; ++c) jt[c] = Jet1<1>(th[M::td_index(c)]);
        if (j < NU) ju[j].d[0] = 1.0; else jx[j - NU].d[0] = 1.0;
        disc_map_adj<M, false, Jet1<1>>(jx, ju, jt, lm, xb, ub, (Jet1<1> *)nullptr, sp.h, sp.rk_steps);
        const double ckk = S.ck(k);
        for (int i = 0; i < NW; ++i)
            Hex[(size_t)k * NW * NW + i * NW + j] = fma(ckk, M::hess(false, i, j, th), i < NU ? ub[i].d[0] : xb[i - NU].d[0]);
    }
    for (int e = tid; e < NW * NW; e += NT) Hex[(size_t)N * NW * NW + e] = S.ck(N) * M::hess(true, e / NW, e % NW, th);
    // barrier diagonal from the final (lam, t) of the bound rows (slacks are constants of the mirror, quirk q1)
    for (int e = tid; e < ne; e += NT) {
        const int k = e / NW, i = e - k * NW;
        double d = 0.0;
        if (!S.skipc(k, i))
            for (int sd = 0; sd < 2; ++sd)
                if (S.has(sd, k, i)) d += S.LAM(sd, e) / S.TT(sd, e);
        S.Dg[e] = d;
    }
    __syncthreads();
    auto HsEx = [&](int k, int i, int j) { return Hex[(size_t)k * NW * NW + i * NW + j]; };
    bool okall = true;
    for (int iu = 0; iu < NU; ++iu) {
        for (int e = tid; e < ne; e += NT) S.rt[e] = e == iu ? -1.0 : 0.0;
        __syncthreads();
        if (iu == 0)
            okall = S.template backward<true>(HsEx, S.rt, nullptr);
        else
            S.template backward<false>(HsEx, S.rt, nullptr);
        S.forward(nullptr);
        // mixed second-order contraction y_v' d/dv (nu' dF/dtheta) + y_nu' dF/dtheta for all theta at once: the tangent of the
        // reverse sweep along (y_v, y_nu), one stage per thread
        for (int k = tid; k < N; k += NT) {
            Jet1<1> jx[NX], ju[NU], jt[NTD], lm[NX], xb[NX], ub[NU], tb[NTD];
            for (int c = 0; c < NU; ++c) ju[c] = Jet1<1>(S.U[k * NU + c]), ju[c].d[0] = S.Du[k * NU + c];
            for (int c = 0; c < NX; ++c) {
                jx[c] = Jet1<1>(S.X[k * NX + c]), jx[c].d[0] = S.Dx[k * NX + c];
                lm[c] = Jet1<1>(nu[(k + 1) * NX + c]), lm[c].d[0] = S.Dnu[(k + 1) * NX + c];
            }
            for (int c = 0; c < NTD; ++c) jt[c] = Jet1<1>(th[M::td_index(c)]);
            disc_map_adj<M, true, Jet1<1>>(jx, ju, jt, lm, xb, ub, tb, sp.h, sp.rk_steps);
            for (int d = 0; d < NTD; ++d) term[k * NTD + d] = tb[d].d[0];
        }
        __syncthreads();
        for (int d = tid; d < NTD; d += NT) {
            double acc = 0.0;
            for (int k = 0; k < N; ++k) acc += term[k * NTD + d];
            dpi[(size_t)iu * NP + M::td_index(d)] = okall ? -acc : NAN;
        }
        for (int e = tid; e < NX * NX + NU * NU; e += NT) {   // y' d2 l / dv dQ_ij = 1/2 (y_i e_j + y_j e_i)
            double acc = 0.0;
            if (e < NX * NX) {
                const int j = e / NX, i = e - j * NX;
                for (int k = 0; k <= N; ++k)
                    acc += 0.5 * S.ck(k) * (S.Dx[k * NX + i] * (S.X[k * NX + j] - xs[j]) + S.Dx[k * NX + j] * (S.X[k * NX + i] - xs[i]));
                dpi[(size_t)iu * NP + M::OFF_Q + e] = okall ? -acc : NAN;
            } else {
                const int ee = e - NX * NX, j = ee / NU, i = ee - j * NU;
                for (int k = 0; k < N; ++k)
                    acc += 0.5 * S.ck(k) * (S.Du[k * NU + i] * S.U[k * NU + j] + S.Du[k * NU + j] * S.U[k * NU + i]);
                dpi[(size_t)iu * NP + M::OFF_R + ee] = okall ? -acc : NAN;
            }
        }
        __syncthreads();
    }
}

}  // namespace mpcrl
