// chain_sens.hpp — sensitivities of the chain solve (dV/dp = dL/dp, nlp.py:1211,1401; du0*/dp by an adjoint Riccati solve,
// nlp.py:1413-1424) as a short sequence of launches on the final iterate (launch_large, mpcrl_api.hip):
//   chain_point_kernel<true>  wavefront per instance: second-order point pass — per evaluation point and link the geometry, the link
//                             Hessians G, the force adjoints q and velocity differences (chain_linearise.hpp)
//   chain_sens_th2            lane per (instance, stage): grad_theta (nu_{k+1}' F_k) read off those tables
//   chain_sens_ad             wavefront per stage group: exact Lagrangian Hessian blocks (tangents x G on the matrix cores)
//   chain_sens_riccati        wavefront per instance: factorisation with the exact Hessian + barrier diagonal, nu adjoint solves
//   chain_sens_mix2           lane per (instance, stage, control): the mixed term y_v' d/dv (nu' dF/dtheta) + y_nu' dF/dtheta
//   chain_sens_out            workgroup per instance: the output reductions
// They re-use [B A], lam, t, nu left in the workspace by the last SQP round (its linearisation is at the final iterate).
#pragma once
#include "chain_linearise.hpp"

namespace mpcrl {

// grad_theta (nu_{k+1}' F_k) on the POINT TABLES (chain_point_kernel<M, true> has walked the adjoint of the RK4 map and left, per
// evaluation point and link, the geometry and the force adjoint q_{e,i}): the parameter adjoint of ode_adj_p written out in those
// quantities — the values of which chain_sens_mix2_kernel forms the tangents.  One lane per (instance, stage), no re-evaluation of the
// map, nothing spilled (until round 5 requests of dV/dp alone ran a reverse sweep of the whole map as one body instead: 459 / 1 243
// spilled registers at n_mass 5 / 7, and a dV/dp that differed by 1e-11 from the one computed next to du0*/dp; now ONE evaluation
// order whatever the flags).  The adjoint of the accelerations that the disturbance gradient needs is the running sum of the q_{e,i}
// from the last link down.
template <class M>
__global__ void __launch_bounds__(64) chain_sens_th2_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NTD = M::NTD, NL = M::NL, TAB2 = M::TAB2, MM = M::M;
    const int N = sp.N;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    const int inst = (int)(gid / N);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const int k = (int)(gid - (long)inst * N);
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double *tab = w + lay.ptab + (size_t)k * 8 * NL * TAB2, *qv = w + lay.qvtab + (size_t)k * 8 * NL * 6;
    const double *mp = th, *Dp = th + NL, *Lp = th + 4 * NL;
    double thb[NTD];
#pragma unroll
    for (int d = 0; d < NTD; ++d) thb[d] = 0.0;
    const int ne = 4 * sp.rk_steps;
#pragma unroll 1
    for (int e = 0; e < ne; ++e) {
        double accb[3] = {0.0, 0.0, 0.0};      // adjoint of the acceleration of mass i - 1 while link i is visited (i = NL - 1 .. 1)
#pragma unroll
        for (int i = NL - 1; i >= 0; --i) {
            const double *t = tab + ((size_t)e * NL + i) * TAB2, *q = qv + ((size_t)e * NL + i) * 6, *dv = q + 3;
            const double inrm = sqrt(t[12] * (1.0 / 3.0)), im = 1.0 / mp[i];
            double thm = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double Lj = Lp[3 * i + j], dm = Dp[3 * i + j] * im;
                const double g = 1.0 - Lj * inrm, fd = q[j] * t[j];
                const double gd = im * (fd * g);
                thb[7 * NL + 3 * i + j] += q[j] * dv[j];
                thb[NL + 3 * i + j] += gd;
                thm -= dm * gd;
                thb[4 * NL + 3 * i + j] -= dm * (fd * inrm);
                if (i > 0) {   // q_i = accb_{i-1} - accb_i  (accb_M = 0: the last link ends at the driven mass)
                    accb[j] = q[j] + (i < MM ? accb[j] : 0.0);
                    thb[10 * NL + 3 * (i - 1) + j] += accb[j];
                }
            }
            thb[i] += thm;
        }
    }
    double *term = w + lay.term + (size_t)k * NTD;
#pragma unroll
    for (int d = 0; d < NTD; ++d) term[d] = thb[d];
}

// Exact Lagrangian Hessian of a stage, Hex_k = c_k hess l_k + hess (nu_{k+1}' F_k)(x_k, u_k), ONE WAVEFRONT PER (instance, stage).
// F is the composition of 4 x rk_steps evaluations of the ODE with linear combinations, and the ODE is nonlinear only through the
// spring forces of the links, so the second-order chain rule collapses to
//     hess (nu' F) = sum over evaluation points e and links i of   (d dist_{e,i} / dv)'  G_{e,i}  (d dist_{e,i} / dv),
// G_{e,i} = the 3 x 3 Hessian of (adjoint of the link force at e)' Fs(dist) (ChainDev::link_hessian), d dist / dv = the first-order
// tangents of the link vectors.  Three passes over the evaluation points, each with a quarter of the live state a forward-over-
// reverse jet sweep needs (which spilled ~1000 registers per lane and was bound by its own scratch traffic):
//   1. the point: RK4 in plain doubles; per-link coefficients of every evaluation point        (chain_point_kernel, one lane per stage)
//   2. the adjoint: reverse sweep of nu_{k+1} through the same points; G_{e,i}                    (chain_point_kernel)
//   3. the tangents: lane j < NW carries direction e_j forward; at every evaluation point the wave publishes Y = d dist / dv
//      (3 NL x NW) and W = G Y and accumulates  Hex += Y' W  on the matrix cores: v_mfma_f64_16x16x4, lower tile triangle, operands
//      straight out of LDS in their register layout (A(i, k) and B(k, j) both at lane 16 k + i|j: measured,
//      profiles/microbench/mfma_f64_16x16x4_probe.hip), results D[r](4 r + lane / 16, lane % 16) stored row-coalesced.
#ifndef MPCRL_CHAIN_AD_ACC_LDS
#define MPCRL_CHAIN_AD_ACC_LDS 1
#endif
template <class M>
struct HexCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB2 = M::TAB2, EV = 8;
    static constexpr int NTI = (NW + 15) / 16;                 // 16-wide tiles per side
    static constexpr int R = 3 * NL, RP = (R + 3) / 4 * 4;      // rows of Y / W per evaluation point, padded to whole k-steps
    static constexpr int LD = 48;                               // row stride of Y / W (doubles): 384 B, so that the two 32-lane halves
                                                                // of a 64-bit LDS read fall on disjoint bank sets
    static constexpr int oTab = 0, oG = oTab + EV * NL * TAB2, oY = (oG + EV * NL * 6 + 1) & ~1, oW = oY + RP * LD, TOTAL = oW + RP * LD;
    static_assert(NTI * 16 <= LD && NW <= 64, "one direction per lane, tiles inside the padded row");
    // STAGES PER WAVEFRONT (round 4): the tangent propagation keeps NW of the 64 lanes busy and runs on LDS latency (this kernel is
    // one wavefront per SIMD: 4 NX doubles of tangents per lane), so a wavefront takes GW-lane groups of consecutive stages — 2 at
    // n_mass 4-6, 4 at n_mass 3: the same wavefront time for 2 (4) stages.  Each group has its own tables, Y / W and accumulators.
    static constexpr int GW = NW <= 16 ? 16 : (NW <= 32 ? 32 : 64), SPW = 64 / GW;
    __host__ __device__ static constexpr int groups(int N) { return (N + SPW - 1) / SPW; }
    static_assert(SPW * TOTAL * 8 <= 40 * 1024, "four wavefronts per CU");
};

template <class M>
__global__ void __launch_bounds__(64, 1) chain_sens_ad_kernel(const LargeSpec sp, const LargeArgs a) {
    using HC = HexCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB2 = M::TAB2, LD = HC::LD, RP = HC::RP, NTI = HC::NTI;
    constexpr int GW = HC::GW, SPW = HC::SPW, NTT = NTI * (NTI + 1) / 2;
    typedef double d4_t __attribute__((ext_vector_type(4)));
    // (n_mass 7 — one stage per wavefront — keeps the lanes' RK4 accumulator in LDS, [index][lane]: with all four NX-vectors of the tangent
    // propagation in registers the kernel spilled 109 of them)
    constexpr bool ACC_LDS = MPCRL_CHAIN_AD_ACC_LDS != 0 && SPW == 1 && (HC::TOTAL + 64 * NX) * 8 <= 40 * 1024;
    __shared__ __attribute__((aligned(16))) double lds[SPW * HC::TOTAL + (ACC_LDS ? 64 * NX : 0)];
    const int N = sp.N, lane = threadIdx.x, NG = HC::groups(N);
    const int inst = blockIdx.x / NG, k0 = (blockIdx.x - inst * NG) * SPW;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double h = sp.h;
    const int steps = sp.rk_steps;
    const int sub = lane / GW, dl = lane - sub * GW;      // this lane's stage group and direction
    double *const mine = lds + sub * HC::TOTAL;
    const double *tab = mine + HC::oTab;
    for (int s_ = 0; s_ < SPW; ++s_) {   // 1. + 2. the point and the adjoint were done per stage by chain_point_kernel<M, true>: its tables -> LDS
        const int k = min(k0 + s_, N - 1);   // (a group past the horizon repeats the last stage and is not stored)
        const double *pt = w + lay.ptab + (size_t)k * HC::EV * NL * TAB2, *gt = w + lay.gtab + (size_t)k * HC::EV * NL * 6;
        double *tb_ = lds + s_ * HC::TOTAL;
        for (int e = lane; e < HC::EV * NL * TAB2; e += 64) tb_[HC::oTab + e] = pt[e];
        for (int e = lane; e < HC::EV * NL * 6; e += 64) tb_[HC::oG + e] = gt[e];
        for (int r = lane; r < RP * LD; r += 64) tb_[HC::oY + r] = 0.0, tb_[HC::oW + r] = 0.0;   // padding rows / columns stay zero
    }
    wave_sync();
    // 3. the tangents and the Hessian accumulation
    d4_t D[SPW][NTT];
#pragma unroll
    for (int s_ = 0; s_ < SPW; ++s_)
#pragma unroll
        for (int t_ = 0; t_ < NTT; ++t_) D[s_][t_] = d4_t{0.0, 0.0, 0.0, 0.0};
    {
        struct AccReg {
            double a[NX];
            MPCRL_DI double &operator[](int i) { return a[i]; }
        };
        struct AccLds {
            double *p;
            MPCRL_DI double &operator[](int i) const { return p[i * 64]; }
        };
        std::conditional_t<ACC_LDS, AccLds, AccReg> acc;
        if constexpr (ACC_LDS) acc.p = lds + SPW * HC::TOTAL + lane;
        double dxc[NX], dk[NX], dxt[NX], du[NU], dd[3 * NL];
#pragma unroll
        for (int i = 0; i < NU; ++i) du[i] = dl == i ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) dxc[i] = dl == NU + i ? 1.0 : 0.0;      // lanes >= NW of a group carry the zero direction
        const int lr = lane >> 4, lc = lane & 15;
        auto accumulate = [&](int e) {   // publish this evaluation point's Y, W = G Y and add Y' W to the tiles
            const double *G = mine + HC::oG + (size_t)e * NL * 6;
            double *Y = mine + HC::oY, *W = mine + HC::oW;
            if (dl < NW) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const double *g = G + i * 6, d0 = dd[3 * i], d1 = dd[3 * i + 1], d2 = dd[3 * i + 2];
                    Y[(3 * i) * LD + dl] = d0, Y[(3 * i + 1) * LD + dl] = d1, Y[(3 * i + 2) * LD + dl] = d2;
                    W[(3 * i) * LD + dl] = g[0] * d0 + g[1] * d1 + g[3] * d2;
                    W[(3 * i + 1) * LD + dl] = g[1] * d0 + g[2] * d1 + g[4] * d2;
                    W[(3 * i + 2) * LD + dl] = g[3] * d0 + g[4] * d1 + g[5] * d2;
                }
            }
            wave_sync();
#pragma unroll
            for (int s_ = 0; s_ < SPW; ++s_) {
                const double *Ys = lds + s_ * HC::TOTAL + HC::oY, *Ws = lds + s_ * HC::TOTAL + HC::oW;
#pragma unroll
                for (int ks = 0; ks < RP / 4; ++ks) {
                    double ya[NTI], wb[NTI];
#pragma unroll
                    for (int t_ = 0; t_ < NTI; ++t_) ya[t_] = Ys[(4 * ks + lr) * LD + 16 * t_ + lc], wb[t_] = Ws[(4 * ks + lr) * LD + 16 * t_ + lc];
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj)
                            D[s_][ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ti], wb[tj], D[s_][ti * (ti + 1) / 2 + tj], 0, 0, 0);
                }
            }
            wave_sync();
        };
        // the tangent of the ODE link by link, the coefficients of the next link in flight while this one is computed (as in
        // chain_dir_pass: one wavefront per SIMD, nothing else covers an LDS round trip)
        constexpr int Mm = M::M;
        double tq[2][12];
        double Cr[3 * NL];
#pragma unroll
        for (int i = 0; i < 3 * NL; ++i) Cr[i] = th[7 * NL + i];
        auto fetch = [&](const double *src, double (&t)[12]) {
#pragma unroll
            for (int j = 0; j < 9; ++j) t[j] = src[j];
        };
        auto eval = [&](const double *tb, const double *dxe, auto par0) {
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[3 * (Mm + 1) + i] = 0.0;
            static_for<NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, cur = (decltype(par0)::value + i) & 1;
                fetch(tb + (i + 1) * TAB2, tq[cur ^ 1]);     // (i + 1 = NL: link 0 of the next evaluation point, the tables are contiguous)
#pragma unroll
                for (int j = 0; j < 3; ++j) tq[cur][9 + j] = Cr[3 * i + j];
                __builtin_amdgcn_sched_barrier(0);
                M::template ode_tan_link<i, true>(tq[cur], dxe, du, dk + 3 * (Mm + 1), dd);
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[i] = dxe[3 * (Mm + 1) + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) dk[3 * Mm + j] = du[j];
        };
        constexpr int P1 = NL & 1, P2 = (2 * NL) & 1, P3 = (3 * NL) & 1;
        fetch(tab, tq[0]);
        for (int s = 0; s < steps; ++s) {
            const double *tb = tab + (size_t)(4 * s) * NL * TAB2;
            eval(tb, dxc, std::integral_constant<int, 0>{});
            accumulate(4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + NL * TAB2, dxt, std::integral_constant<int, P1>{});
            accumulate(4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + 2 * NL * TAB2, dxt, std::integral_constant<int, P2>{});
            accumulate(4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + h * dk[i];
            eval(tb + 3 * NL * TAB2, dxt, std::integral_constant<int, P3>{});
            accumulate(4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) dxc[i] = dxc[i] + (h / 6.0) * (acc[i] + dk[i]);
        }
        // Hex_k = c_k hess l_k + the accumulated second-order term; D[.][r] holds (row 4 r + lane / 16, column lane % 16) of its tile
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            const int k = k0 + s_;
            if (k >= N) break;
            const double ckk = sp.cost_kind == 0 ? sp.dT : (k == 0 ? sp.dT : pow(sp.gamma, (double)k) * sp.dT);
            double *Hex = w + lay.Hex + (size_t)k * NW * NW;
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ti + 4 * r + lr, j = 16 * tj + lc;
                        if (i < NW && j < NW) {
                            const double v = fma(ckk, M::hess(false, i, j, th), D[s_][ti * (ti + 1) / 2 + tj][r]);
                            Hex[i * NW + j] = v;
                            if (ti != tj) Hex[j * NW + i] = v;
                        }
                    }
        }
    }
}

template <class M>
__global__ void __launch_bounds__(64, 1) chain_sens_riccati_kernel(const LargeSpec sp, const LargeArgs a) {
    using Cfg = ChainCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // Cfg::lds_doubles(N) doubles (launch_large)
    __shared__ int sidx[196];
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    ChainSolver<M> S(sp, lane);
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    S.bind_workspace(w, lay);
    S.setup(lds, sidx);
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    const int ne = (N + 1) * NW;
    const double *th = S.th, *xs = sp.consts;
    const WsArr Hex{(char *)w, (unsigned)lay.Hex};
    if (!((a.flags & 2) && a.dpi) || S.qmode) return;
    for (int e = lane; e < NW * NW; e += NT) Hex[N * NW * NW + e] = S.ck(N) * M::hess(true, e / NW, e % NW, th);
    // barrier diagonal from the final (lam, t) of the bound rows (slacks are constants of the mirror, quirk q1)
    for (int e = lane; e < ne; e += NT) {
        const int k = e / NW, i = e - k * NW;
        double d = 0.0;
        if (!S.skipc(k, i))
            for (int sd = 0; sd < 2; ++sd)
                if (S.has(sd, k, i)) d += S.LAM(sd, e) / S.TT(sd, e);
        S.Dg[e] = d;
    }
    wave_sync();
    for (int e = lane; e < N * NX; e += NT) S.rb[e] = 0.0;   // no dynamics offset in the adjoint systems
    const WsArr Ydx{(char *)w, (unsigned)lay.Ydx}, Ydu{(char *)w, (unsigned)lay.Ydu}, Ydnu{(char *)w, (unsigned)lay.Ydnu};
    // one factor sweep (P_k streamed), ONE forward sweep for the NU adjoint solves
    for (int e = lane; e < ne; e += NT) S.rt[e] = e == 0 ? -1.0 : 0.0;
    wave_sync();
    const bool okall = ChainSolver<M>::template factor_call<HessGlobal<M>>(S.ctx(), Hex.off, S.rt.off, S.rb.off);
    ChainSolver<M>::forward_sens_call(S.ctx(), Ydx.off, Ydu.off, Ydnu.off);
    if (lane == 0) S.state[ST_STATUS] = okall ? 0.0 : 4.0;   // read by sens_out: NaN sensitivities when the exact-Hessian KKT matrix is not pd
}

// ---- the mixed term on the POINT TABLES (round 4; until then a forward-over-reverse jet sweep of the whole
// RK4 map that kept four stage states, four adjoint vectors and the parameter adjoint live as jets: 1 076 spilled registers and 3.3 KB of
// scratch per lane at n_mass 5, 1 696 / 6 KB at n_mass 7 — 0.43 / 1.85 ms for 12 k flops per lane).  Per (instance, stage, control)
//     term2 = d/d eps  grad_theta [nu' F](v + eps y_v, nu + eps y_nu)
// F is RK4 steps of an ODE whose only nonlinearity is the spring force of each link, so everything second order is local to a
// (evaluation point e, link i): with the point tables of chain_point_pass<true> (dist, spring coefficients; G_{e,i}; the force
// adjoint q_{e,i} and the velocity difference dv_{e,i}) the sweep is
//   1. the tangent of the evaluation states along y_v: plain ode_tan on the tables (what the direction pass does), one RK4 step at a time;
//   2. the tangent of the reverse sweep: the same linear recursion as the values (ode_tan_T) on the tangent adjoints, plus the source
//      (d dist / dx)' G_{e,i} d dist_{e,i} at every evaluation point;
//   3. at every (e, i) the tangent of the parameter adjoint of ode_adj_p, written out in the table quantities.
// ~170 doubles of live state, no jets, no scratch.
template <class M>
__global__ void __launch_bounds__(64) chain_sens_mix2_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD, NL = M::NL, TAB2 = M::TAB2, MM = M::M;
    // per lane in LDS ([index][lane]: conflict-free): the NTD accumulators of the parameter-adjoint tangent and the start state of
    // the second RK4 step — with the evaluation states recomputed where they are used, what stays in registers is one evaluation
    // state, one ODE tangent and the four adjoint vectors
    extern __shared__ double sm[];
    const int N = sp.N, lane = threadIdx.x;
    double *thd = sm + lane, *dx1 = sm + NTD * 64 + lane;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    const int per = N * NU;
    const int inst = (int)(gid / per);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    // (the control index runs fastest: the NU lanes of a stage read the SAME table entries — every table access of this kernel is a gather,
    // one cache line per lane, and the kernel runs on the rate of those requests; with the stage index fastest all 64 lanes of a load
    // went to different lines)
    const int it = (int)(gid - (long)inst * per), k = it / NU, iu = it - k * NU;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double *tab = w + lay.ptab + (size_t)k * 8 * NL * TAB2, *Gt = w + lay.gtab + (size_t)k * 8 * NL * 6, *qv = w + lay.qvtab + (size_t)k * 8 * NL * 6;
    const double *Ydx = w + lay.Ydx + (size_t)iu * (N + 1) * NX + (size_t)k * NX, *Ydu = w + lay.Ydu + (size_t)iu * N * NU,
                 *Ydnu = w + lay.Ydnu + (size_t)iu * (N + 1) * NX;
    const double h = sp.h;
    const int steps = sp.rk_steps;
    const double *mp = th, *Dp = th + NL, *Lp = th + 4 * NL;
    double du[NU], lbd[NX];
#pragma unroll
    for (int c = 0; c < NU; ++c) du[c] = Ydu[k * NU + c];
#pragma unroll
    for (int c = 0; c < NX; ++c) lbd[c] = Ydnu[(k + 1) * NX + c];
#pragma unroll
    for (int d = 0; d < NTD; ++d) thd[d * 64] = 0.0;
    auto dx0 = [&](int s, int i) { return s == 0 ? Ydx[i] : dx1[i * 64]; };      // start state of step s
    // evaluation state n (0..3) of step s into dX.  Runtime loops on purpose (no unrolling over the evaluation points): straight-line
    // code over the eight points lets the scheduler stretch live ranges over all of them, and the kernel is back in scratch
    auto state_at = [&](int s, int n, double (&dX)[NX]) {
        const double *tb = tab + (size_t)(4 * s) * NL * TAB2;
#pragma unroll
        for (int i = 0; i < NX; ++i) dX[i] = dx0(s, i);
#pragma unroll 1
        for (int m = 0; m < n; ++m) {
            double dk[NX];
            M::template ode_tan<TAB2, false>(tb + (size_t)m * NL * TAB2, th, dX, du, dk, nullptr);
            const double c = m == 2 ? h : 0.5 * h;
#pragma unroll
            for (int i = 0; i < NX; ++i) dX[i] = dx0(s, i) + c * dk[i];
        }
    };
    if (steps > 1) {      // start state of the second step: one full RK4 step of the tangent
        const double *tb = tab;
        double dX[NX], acc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dX[i] = dx0(0, i), acc[i] = 0.0;
#pragma unroll 1
        for (int m = 0; m < 4; ++m) {
            double dk[NX];
            M::template ode_tan<TAB2, false>(tb + (size_t)m * NL * TAB2, th, dX, du, dk, nullptr);
            const double c = m == 2 ? h : 0.5 * h, wgt = (m == 0 || m == 3) ? 1.0 : 2.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = fma(wgt, dk[i], acc[i]), dX[i] = dx0(0, i) + c * dk[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) dx1[i * 64] = dx0(0, i) + (h / 6.0) * acc[i];
    }
#pragma unroll 1
    for (int s = steps - 1; s >= 0; --s) {
        double kbd[NX], Xbd[NX], accd[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) accd[i] = lbd[i], Xbd[i] = 0.0;
#pragma unroll 1
        for (int n = 3; n >= 0; --n) {      // evaluation point e = 4 s + n: Xbd = J' kbd + Hessian source; parameter-adjoint tangents
            // weights of the reverse RK4 sweep: kb_3 = h/6 lb, kb_2 = h/3 lb + h Xb_3, kb_1 = h/3 lb + h/2 Xb_2, kb_0 = h/6 lb + h/2 Xb_1
            const double ca = (n == 3 || n == 0) ? h / 6.0 : h / 3.0, cb = n == 3 ? 0.0 : (n == 2 ? h : 0.5 * h);
#pragma unroll
            for (int i = 0; i < NX; ++i) kbd[i] = ca * lbd[i] + cb * Xbd[i];
            const int e = 4 * s + n;
            const double *tb = tab + (size_t)e * NL * TAB2;
            double qd[3 * NL], dX[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) Xbd[i] = 0.0;
            M::template ode_tan_T<TAB2>(tb, th, kbd, Xbd, qd);
#pragma unroll
            for (int i = 0; i < 3 * MM; ++i) thd[(10 * NL + i) * 64] += kbd[3 * (MM + 1) + i];        // w enters the accelerations directly
            state_at(s, n, dX);
            const double *dpos = dX, *dvel = dX + 3 * (MM + 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const double *t = tb + i * TAB2, *G = Gt + ((size_t)e * NL + i) * 6, *q = qv + ((size_t)e * NL + i) * 6, *dv = q + 3;
                double dd[3], ddv[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    dd[j] = i ? dpos[3 * i + j] - dpos[3 * (i > 0 ? i - 1 : 0) + j] : dpos[j];
                    const double dvr = i < MM ? dvel[3 * (i < MM ? i : 0) + j] : du[j];
                    ddv[j] = i ? dvr - dvel[3 * (i > 0 ? i - 1 : 0) + j] : dvr;
                }
                const double w0 = G[0] * dd[0] + G[1] * dd[1] + G[3] * dd[2], w1 = G[1] * dd[0] + G[2] * dd[1] + G[4] * dd[2],
                             w2 = G[3] * dd[0] + G[4] * dd[1] + G[5] * dd[2];
                const double wv[3] = {w0, w1, w2};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Xbd[3 * i + j] += wv[j];
                    if (i > 0) Xbd[3 * (i > 0 ? i - 1 : 0) + j] -= wv[j];
                }
                const double inrm = sqrt(t[12] * (1.0 / 3.0)), sdot = t[0] * dd[0] + t[1] * dd[1] + t[2] * dd[2];
                const double dinrm = -(inrm * inrm * inrm) * sdot, im = 1.0 / mp[i];
                double thm = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double Lj = Lp[3 * i + j], dm = Dp[3 * i + j] * im;
                    const double g = 1.0 - Lj * inrm, dg = -Lj * dinrm;
                    const double fd = q[j] * t[j], dfd = qd[3 * i + j] * t[j] + q[j] * dd[j];
                    const double dgd = im * (dfd * g + fd * dg);
                    thd[(7 * NL + 3 * i + j) * 64] += qd[3 * i + j] * dv[j] + q[j] * ddv[j];
                    thd[(NL + 3 * i + j) * 64] += dgd;
                    thm -= dm * dgd;
                    thd[(4 * NL + 3 * i + j) * 64] -= dm * (dfd * inrm + fd * dinrm);
                }
                thd[i * 64] += thm;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) accd[i] += Xbd[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) lbd[i] = accd[i];
    }
    double *term2 = w + lay.term2 + ((size_t)iu * N + k) * NTD;
#pragma unroll
    for (int d = 0; d < NTD; ++d) term2[d] = thd[d * 64];
}

// One output element per lane: slot 0 = dV/dp (with MPCRL_SENS_V), slots 1..NU = rows of du0*/dp (with MPCRL_SENS_PI).
// Each element is a sum over the stages of per-stage terms left in the workspace, or of closed forms in (X, U, adjoint solution).
template <class M>
__global__ void __launch_bounds__(1024) chain_sens_out_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD, NP = M::NP;
    constexpr int NE = NTD + NX * NX + NU * NU;   // elements per slot: dynamics parameters, Q (column-major), R
    constexpr int PER = (NU + 1) * NE;
    const int N = sp.N;
    // One workgroup of 1024 lanes per INSTANCE.  The Q / R outputs are sums over the stages of products of two trajectory entries:
    // the trajectories (X - x_ss, U, the NU adjoint solutions: 32 KB at n_mass 5) are staged in LDS once, coalesced, and every lane
    // then takes outputs rem = lane, lane + 1024, ...  (Round 3 ran one lane per output on 256-lane workgroups that each went to
    // global memory with 64 different addresses per load instruction: 0.22 ms; staged per 256 outputs: 0.20 ms — the staging latency
    // of 8 workgroups per instance, 6 rounds of them on the chip, was the time.)
    extern __shared__ double sm[];
    double *cks = sm, *lX = sm + 64, *lU = lX + (N + 1) * NX, *lY = lU + N * NU;   // lY: [NU][(N+1) NX + N NU]
    const int inst = blockIdx.x, tid = threadIdx.x;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const bool want_v = (a.flags & 1) && a.dV, want_pi = (a.flags & 2) && a.dpi && !a.u0fix;
    const LargeLayout<M> lay(N);
    const double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU, *xs = sp.consts;
    const int ny = (N + 1) * NX + N * NU;
    for (int k = tid; k <= N; k += 1024) {
        double c = k == N ? 1.0 : sp.dT;
        if (sp.cost_kind != 0) c = k == 0 ? sp.dT : (k == N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);
        cks[k] = c;
    }
    for (int e = tid; e < (N + 1) * NX; e += 1024) lX[e] = X[e] - xs[e % NX];
    for (int e = tid; e < N * NU; e += 1024) lU[e] = U[e];
    if (want_pi)
        for (int e = tid; e < NU * ny; e += 1024) {
            const int iu = e / ny, o = e - iu * ny;
            lY[e] = o < (N + 1) * NX ? w[lay.Ydx + (size_t)iu * (N + 1) * NX + o] : w[lay.Ydu + (size_t)iu * N * NU + (o - (N + 1) * NX)];
        }
    __syncthreads();
    const bool sens_ok = w[lay.state + ST_STATUS] == 0.0;
    for (int rem = tid; rem < PER; rem += 1024) {
        const int slot = rem / NE, e0 = rem - slot * NE;
        if (slot == 0 ? !want_v : !want_pi) continue;
        const int iu = slot - 1;
        const double *tm = slot == 0 ? w + lay.term : w + lay.term2 + (size_t)iu * N * NTD;
        const double *Dx = lY + (iu < 0 ? 0 : iu) * ny, *Du = Dx + (N + 1) * NX;
        double acc = 0.0;
        int pidx;
        if (e0 < NTD) {
            for (int k = 0; k < N; ++k) acc += tm[k * NTD + e0];
            pidx = M::td_index(e0);
        } else if (e0 < NTD + NX * NX) {
            const int e = e0 - NTD, j = e / NX, i = e - j * NX;   // column-major position of Q(i, j)
            if (slot == 0) {   // d/dQ_ij of sum_k c_k l_k (ocp_utils.py:276-277)
                for (int k = 0; k <= N; ++k) acc = fma(0.5 * cks[k] * lX[k * NX + i], lX[k * NX + j], acc);
            } else {           // y' d2 l / dv dQ_ij = 1/2 (y_i e_j + y_j e_i)
                for (int k = 0; k <= N; ++k) acc += 0.5 * cks[k] * (Dx[k * NX + i] * lX[k * NX + j] + Dx[k * NX + j] * lX[k * NX + i]);
            }
            pidx = M::OFF_Q + e;
        } else {
            const int ee = e0 - NTD - NX * NX, j = ee / NU, i = ee - j * NU;
            if (slot == 0) {
                for (int k = 0; k < N; ++k) acc = fma(0.5 * cks[k] * lU[k * NU + i], lU[k * NU + j], acc);
            } else {
                for (int k = 0; k < N; ++k) acc += 0.5 * cks[k] * (Du[k * NU + i] * lU[k * NU + j] + Du[k * NU + j] * lU[k * NU + i]);
            }
            pidx = M::OFF_R + ee;
        }
        if (slot == 0)
            a.dV[(size_t)inst * NP + pidx] = acc;
        else
            a.dpi[((size_t)inst * NU + iu) * NP + pidx] = sens_ok ? -acc : NAN;
    }
}

}  // namespace mpcrl
