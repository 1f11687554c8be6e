// chain_linearise.hpp — the SQP loop of the chain solver: iterate set-up, the linearisation of the 2-step RK4 map (point pass,
// direction pass) and chain_sqp_kernel, which runs all SQP rounds of an instance on one wavefront in one launch.
#pragma once
#include "chain_sweeps.hpp"

namespace mpcrl {

// ---- iterate set-up: cold start (MPC.reset, mpc.py:204-210) or the stored one.  One wavefront per instance.
template <class M>
__global__ void __launch_bounds__(64) chain_init_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NT = 64;
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU;
    const double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    const double *bnd = a.BND + (size_t)inst * 10 * nb;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = a.u0fix ? a.u0fix + (size_t)inst * NU : nullptr;
    double *NUv = w + lay.ynu, *lam = w + lay.lamw, *t = w + lay.tw, *aff = w + lay.aff, *st = w + lay.state;
    const int ne = (N + 1) * NW;
    double stepn = -1.0;   // perturbation seen by the first QP (< 0: cold)
    if ((a.flags & 8) || (a.cold && a.cold[inst])) {
        for (int e = lane; e < (N + 1) * NX; e += NT) X[e] = x0[e % NX], NUv[e] = 0.0;
        for (int e = lane; e < N * NU; e += NT) U[e] = 0.0;
        for (int e = lane; e < 2 * ne; e += NT) lam[e] = 0.0, t[e] = 1.0, aff[e] = 0.0;
    } else {
        for (int e = lane; e < (N + 1) * NX; e += NT) NUv[e] = e < NX ? 0.0 : PIg[e - NX];
        for (int e = lane; e < 2 * ne; e += NT) lam[e] = bnd[e], t[e] = bnd[2 * nb + e], aff[e] = 0.0;
        double sl = 0.0;
        if (lane < NX) sl = fabs(x0[lane] - X[lane]);
        if (u0f && lane < NU) sl = fmax(sl, fabs(u0f[lane] - U[lane]));
        stepn = (a.flags & 16) ? -1.0 : wave_max(sl);   // MPCRL_COLD_DUAL: the interior point starts from its default point
    }
    if (lane == 0) {
        st[ST_ACTIVE] = 1.0, st[ST_IT] = 0.0, st[ST_NIPM] = 0.0, st[ST_TIGHT] = 1.0, st[ST_STEPN] = stepn, st[ST_COST] = 0.0;
        st[ST_STATUS] = 2.0;
        for (int j = 0; j < 4; ++j) st[ST_RES + j] = 0.0;
    }
}

// ---- the POINT pass of the derivative kernels: one lane per (instance, stage) walks the 4 x rk_steps evaluation points of the RK4
// map in plain doubles and leaves, per evaluation point and link, what the tangent of the ODE needs there (ChainDev::ode_coef) in the
// instance's workspace; it also writes r_k = F(x_k, u_k) - x_{k+1}.  With SECOND (sensitivities) the tables carry the second-order
// coefficients as well and a reverse sweep of nu_{k+1} through the same points adds the 3 x 3 Hessian of every link force
// (ChainDev::link_hessian).  The direction kernels below (one lane per direction) then start from these tables: done inside them, this
// pass was repeated by every lane of a stage — 60 % of the linearisation's and half of the Hessian kernel's instructions.
// the pass itself, for stage k of one instance (X, U, th: the instance's iterate and parameters, w: its workspace).  A real call: the
// QP kernel runs it at the end of a round on N of its lanes, with a register allocation of its own.
// TH_LDS (the SQP kernel): th is the instance's COMPACT parameter copy in LDS (the NTD differentiable entries, staged by the caller) —
// out of the parameter vector in global memory every evaluation point fetched its ~50-75 coefficients again, behind the table
// stores of the point before.
template <class M, bool SECOND, bool TH_LDS>
MPCRL_DI void chain_point_body(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_);
template <class M, bool SECOND, bool TH_LDS>
__device__ MPCRL_PHASE_FN void chain_point_pass(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_) {
    chain_point_body<M, SECOND, TH_LDS>(X_, U_, th_, w_, N, k, h, steps, lacc_);
}
template <class M, bool SECOND, bool TH_LDS>
MPCRL_DI void chain_point_body(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_) {
    constexpr int NX = M::NX, NU = M::NU, NL = M::NL, TS = SECOND ? M::TAB2 : M::TAB;
    const double *X = as_global(X_), *U = as_global(U_), *th = TH_LDS ? as_lds(th_) : as_global(th_);
    double *w = as_global(w_);
    const LargeLayout<M> lay(N);
    double *tab = w + lay.ptab + (size_t)k * 8 * NL * M::TAB2;
    double u[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = U[k * NU + i];
    struct AccReg {
        double a[NX];
        MPCRL_DI double &operator[](int i) { return a[i]; }
    };
    struct AccLds {
        double *p;
        int st;
        MPCRL_DI double &operator[](int i) const { return p[i * st]; }
    };
    {
        // (TH_LDS: the RK4 accumulator of the lane lives in LDS, entry i at lacc[i N + k].  With all four arrays in registers the
        // compiler kept ~8 doubles of them in scratch, and every reload — an s_waitcnt vmcnt(0) — also waited for the table stores in
        // flight, 40 scattered lines each: ~20 drains per RK4 step were most of this pass's time.)
        std::conditional_t<TH_LDS, AccLds, AccReg> acc;
        if constexpr (TH_LDS) acc.p = as_lds(lacc_) + k, acc.st = N;
        double xc[NX], kk[NX], xt[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = X[k * NX + i];
        // (with SECOND also the velocity difference of every link at every evaluation point: the mixed term wants it, chain_sens_mix2)
        auto store_dv = [&](const double *xs_, int e) {
            if constexpr (SECOND) {
                double *qv = w + lay.qvtab + ((size_t)k * 8 + e) * NL * 6;
                constexpr int Mm = M::M;
#pragma unroll
                for (int i = 0; i < NL; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double vr = i < Mm ? xs_[3 * (Mm + 1) + 3 * (i < Mm ? i : 0) + j] : u[j];
                        qv[6 * i + 3 + j] = i ? vr - xs_[3 * (Mm + 1) + 3 * (i > 0 ? i - 1 : 0) + j] : vr;
                    }
            }
        };
        for (int s = 0; s < steps; ++s) {
            double *tb = tab + (size_t)(4 * s) * NL * TS;
            M::template ode_coef<SECOND, TH_LDS>(xc, u, th, kk, tb, true);
            store_dv(xc, 4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + NL * TS, true);
            store_dv(xt, 4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + 2 * NL * TS, true);
            store_dv(xt, 4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + h * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + 3 * NL * TS, true);
            store_dv(xt, 4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xc[i] + (h / 6.0) * (acc[i] + kk[i]);
        }
        if constexpr (!SECOND) {
            double *r = w + lay.r + k * NX;
#pragma unroll
            for (int i = 0; i < NX; ++i) r[i] = xc[i] - X[(k + 1) * NX + i];
        }
    }
    if constexpr (SECOND) {   // the adjoint: kb_e = d(nu' F) / d(k_e) at every evaluation point, last step first; G_{e,i} from its force part
        const double *nu = w + lay.ynu;
        double *Gt = w + lay.gtab + (size_t)k * 8 * NL * 6;
        std::conditional_t<TH_LDS, AccLds, AccReg> acc;
        if constexpr (TH_LDS) acc.p = as_lds(lacc_) + k, acc.st = N;
        double lb[NX], kb[NX], Xb[NX], q[3 * NL];
#pragma unroll
        for (int i = 0; i < NX; ++i) lb[i] = nu[(k + 1) * NX + i];
        for (int s = steps - 1; s >= 0; --s) {
            auto node = [&](int e) {   // Xb = J(e)' kb, and the link Hessians of this evaluation point
                const double *tb = tab + (size_t)e * NL * TS;
#pragma unroll
                for (int i = 0; i < NX; ++i) Xb[i] = 0.0;
                M::template ode_tan_T<TS>(tb, th, kb, Xb, q);
                double *qv = w + lay.qvtab + ((size_t)k * 8 + e) * NL * 6;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    M::link_hessian(tb + i * TS, q + 3 * i, Gt + ((size_t)e * NL + i) * 6);
#pragma unroll
                    for (int j = 0; j < 3; ++j) qv[6 * i + j] = q[3 * i + j];
                }
            };
#pragma unroll
            for (int i = 0; i < NX; ++i) kb[i] = (h / 6.0) * lb[i];
            node(4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = lb[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + h * Xb[i];
            node(4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + (0.5 * h) * Xb[i];
            node(4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 6.0) * lb[i] + (0.5 * h) * Xb[i];
            node(4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) lb[i] = acc[i] + Xb[i];
        }
    }
}

// One wavefront per instance, lane = stage (N <= 64): the instance's differentiable parameters and the lanes' RK4 accumulators sit in LDS
// (as in the SQP kernel's call).  Until round 4 the lanes of a wavefront ran over (instance, stage) pairs with everything in
// registers: ~50 doubles of them in scratch, every reload waiting for the scattered table stores in flight — 261 us at n_mass 5 for a
// pass that takes 35 us inside the SQP kernel.
template <class M, bool SECOND>
__global__ void __launch_bounds__(64) chain_point_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU;
    extern __shared__ __attribute__((aligned(16))) double lds[];   // NTD + N NX doubles
    const int N = sp.N, inst = blockIdx.x, lane = threadIdx.x;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    if constexpr (SECOND) {
        const int status = a.status[inst];
        if (!(status == 0 || status == 2)) return;
    } else {
        if (w[lay.state + ST_ACTIVE] == 0.0) return;
    }
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    for (int e = lane; e < M::NTD; e += 64) lds[e] = th[M::td_index(e)];
    wave_sync();
    if (lane < N)
        chain_point_pass<M, SECOND, true>(a.X + (size_t)inst * (N + 1) * NX, a.U + (size_t)inst * N * NU, lds, w, N, lane, sp.h, sp.rk_steps, lds + M::NTD);
}

// ---- dynamics linearisation, the DIRECTION pass: fills [B A]_k of all stages of ONE instance, run by the instance's own wavefront
// inside the SQP kernel.  One lane per (stage, direction), 64 of them per step; the coefficient tables (chain_point_pass) of the
// stages a step touches are copied to LDS, then each lane propagates ONLY its tangent through the evaluation points.  A forward jet
// per lane (value + tangent through the whole map) needs both sets of arrays live at once — 2 x 4 NX doubles, 528 registers at
// NX = 33, i.e. spills whose scratch traffic made the round-1 kernel HBM-bound (46 GB per step at n_mass = 7) — and recomputes the
// point NW times.  As a grid-wide kernel of its own (first half of round 2) it cost the same SIMD time — an instance's 40 x NW
// directions are 15 (23) wavefront-steps either way, and a batch of 1024 is one wavefront per SIMD — plus a kernel boundary per round.
template <class M>
struct DirCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, EV = 8;
    static constexpr int TSZ = EV * NL * M::TAB;                    // doubles of one stage's first-order table in the workspace
    // In LDS a link's record is 12 doubles — its 9 coefficients and the link's 3 damping coefficients — on a 16-byte boundary: six
    // ds_read_b128 per link (4 LDS cycles each) where the 9 + 3 doubles at odd offsets were five ds_read2_b64 (8 cycles each: the
    // instruction runs at half the LDS rate) and two ds_read_b64.  Four wavefronts per CU run this pass at the same time and it is
    // bound by the LDS pipe they share.
    // (+ 8 doubles between the stages' tables: EV NL LREC doubles is a multiple of the 64 banks for every chain size, so the lanes of
    // two stages in one ds_read_b128 lane group read different addresses on the SAME banks — a 2-way conflict in three of the four
    // groups of a step; 16 dwords apart they do not meet)
    static constexpr int LREC = 12, TSZL = EV * NL * LREC + 8;
    static constexpr int SPAN = (64 + NW - 1) / NW + 1;              // stages a step of 64 consecutive (stage, direction) items can touch
    static constexpr bool FITS = SPAN * TSZL <= 2048;                // else whole stages per step
    static constexpr int NST = FITS ? SPAN : 64 / NW;                // stage tables in LDS
    static constexpr int LP = FITS ? 64 : (64 / NW) * NW;            // items per step
    static constexpr int CO = NST * TSZL;                            // after the tables: the instance's NTD differentiable parameters
#ifndef MPCRL_CHAIN_DIR_ACCL
#define MPCRL_CHAIN_DIR_ACCL 16
#endif
    static constexpr int ACCL = NX > 21 ? MPCRL_CHAIN_DIR_ACCL : 0;   // entries of the lane's RK4 accumulator kept in LDS (64 lanes x ACCL doubles)
    static_assert(ChainCfg<M>::oBig % 2 == 0, "16-byte records");
};

template <class M>
MPCRL_DI void chain_dir_body(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps);
template <class M>
__device__ MPCRL_PHASE_FN void chain_dir_pass(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps) {
    chain_dir_body<M>(th_, w_, tabl_, N, lane, h, steps);
}
template <class M>
MPCRL_DI void chain_dir_body(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps) {
    using DC = DirCfg<M>;
    const double *th = as_global(th_);
    double *w = as_global(w_), *tabl = as_lds(tabl_);
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB = M::TAB, TSZ = DC::TSZ, STG = DC::EV * NL * M::TAB2;
    const LargeLayout<M> lay(N);
    const int items = N * NW;
    // the tables of step i + 1 are requested before step i is computed and go to LDS after it: one global round trip per step hidden
    constexpr int NPF = (DC::NST * TSZ + 63) / 64, LREC = DC::LREC, TSZL = DC::TSZL;
    double pf[NPF];
    auto request = [&](int i0) {
        const int k_lo = i0 / NW, cnt = (min(N - 1, (i0 + DC::LP - 1) / NW) - k_lo + 1) * TSZ;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int e = lane + 64 * j, ks = e / TSZ;
            pf[j] = e < cnt ? w[lay.ptab + (size_t)(k_lo + ks) * STG + (e - ks * TSZ)] : 0.0;
        }
    };
    request(0);
    // the damping coefficients (out of the compact parameter copy the caller staged) go into every link record once: a record's
    // link index is a function of its position
    for (int r = lane; r < DC::NST * DC::EV * NL; r += 64) {
        const int st_ = r / (DC::EV * NL), rr = r - st_ * (DC::EV * NL);
#pragma unroll
        for (int j = 0; j < 3; ++j) tabl[st_ * TSZL + rr * LREC + 9 + j] = tabl[DC::CO + 7 * NL + 3 * (rr % NL) + j];
    }
    for (int i0 = 0; i0 < items; i0 += DC::LP) {
        const int k_lo = i0 / NW, k_hi = min(N - 1, (i0 + DC::LP - 1) / NW);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int e = lane + 64 * j;
            const int ks = e / TSZ, idx = e - ks * TSZ, rec = idx / M::TAB;
            if (e < (k_hi - k_lo + 1) * TSZ) tabl[ks * TSZL + rec * LREC + (idx - rec * M::TAB)] = pf[j];
        }
        wave_sync();
        if (i0 + DC::LP < items) request(i0 + DC::LP);
        const int it_ = i0 + lane;
        const bool on = lane < DC::LP && it_ < items;
        const int k = on ? it_ / NW : k_lo, d = on ? it_ - k * NW : 0;
        const double *mytab = tabl + (size_t)(k - k_lo) * TSZL;
        // (n_mass 6 / 7: the four NX-vectors of the lane overflow the 256 vector registers — AGPR copies inside the loop; the first ACCL
        // entries of the RK4 accumulator live in LDS instead, [index][lane], where the point pass kept its accumulators: dead by now)
        constexpr int ACCL = DC::ACCL;
        struct Acc {
            double r[NX - ACCL > 0 ? NX - ACCL : 1];
            double *p;
            MPCRL_DI double get(int i) const { return i < ACCL ? p[i * 64] : r[i >= ACCL ? i - ACCL : 0]; }
            MPCRL_DI void set(int i, double v) {
                if (i < ACCL) p[i * 64] = v; else r[i >= ACCL ? i - ACCL : 0] = v;
            }
        } acc;
        acc.p = tabl + DC::CO + M::NTD + (ChainCfg<M>::FUSE_GT ? (N + 1) * NX : 0) + lane;
        double dxc[NX], dk[NX], dxt[NX], du[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) du[i] = d == i ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) dxc[i] = d == NU + i ? 1.0 : 0.0;
        // The tangent of the ODE at an evaluation point, link by link, with the 9 coefficients of the NEXT link (the next evaluation
        // point's first one after the last; and the link's damping coefficients) requested from LDS before the current link is computed: this wavefront is alone on its
        // SIMD, so nothing else covers the LDS round trip, and with 4 NX doubles of tangents live the compiler placed every read
        // right in front of its use (55 waits on an empty LDS queue per RK4 step: the pass ran on LDS latency).
        constexpr int Mm = M::M;
        double tq[2][12];
        auto fetch = [&](const double *src, double (&t)[12]) {   // a link record: 9 coefficients of the point + the link's damping
            if constexpr (NX <= 21) {
                const d2_t *s2 = (const d2_t *)__builtin_assume_aligned(src, 16);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const d2_t v = s2[j];
                    t[2 * j] = v.x, t[2 * j + 1] = v.y;
                }
            } else {   // (n_mass 6, 7: the lane is out of registers and the aligned register quads of ds_read_b128 cost more than they save: 2.53 vs 2.08 ms)
#pragma unroll
                for (int j = 0; j < 12; ++j) t[j] = src[j];
            }
        };
        auto eval = [&](const double *tb, const double *dxe, auto par0) {   // dk = (d f / d x) dxe + (d f / d u) du; par0: buffer of link 0
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[3 * (Mm + 1) + i] = 0.0;
            static_for<NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, cur = (decltype(par0)::value + i) & 1;
                fetch(tb + (i + 1) * LREC, tq[cur ^ 1]);   // (i + 1 = NL: link 0 of the next evaluation point, the tables are contiguous)
                __builtin_amdgcn_sched_barrier(0);
                M::template ode_tan_link<i>(tq[cur], dxe, du, dk + 3 * (Mm + 1));
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[i] = dxe[3 * (Mm + 1) + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) dk[3 * Mm + j] = du[j];
        };
        constexpr int P1 = NL & 1, P2 = (2 * NL) & 1, P3 = (3 * NL) & 1;   // buffer parity at the start of the evaluation points
        static_assert(((4 * NL) & 1) == 0, "an RK4 step ends on the buffer it started with");
        fetch(mytab, tq[0]);
        for (int s_ = 0; s_ < steps; ++s_) {
            const double *tb = mytab + (size_t)(4 * s_) * NL * LREC;
            eval(tb, dxc, std::integral_constant<int, 0>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc.set(i, dk[i]), dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + NL * LREC, dxt, std::integral_constant<int, P1>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc.set(i, acc.get(i) + 2.0 * dk[i]), dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + 2 * NL * LREC, dxt, std::integral_constant<int, P2>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc.set(i, acc.get(i) + 2.0 * dk[i]), dxt[i] = dxc[i] + h * dk[i];
            eval(tb + 3 * NL * LREC, dxt, std::integral_constant<int, P3>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) dxc[i] = dxc[i] + (h / 6.0) * (acc.get(i) + dk[i]);
        }
        if (on) {
            double *BA = w + lay.BA + (size_t)k * LargeLayout<M>::BAS;
#pragma unroll
            for (int i = 0; i < NX; ++i) BA[i * NW + d] = dxc[i];
            if constexpr (ChainCfg<M>::FUSE_GT) {
                // ([B A]_k' nu_{k+1})_d: this lane holds column d.  What the stationarity residual of the round wants (round_start) —
                // as a stage pass over [B A] there it was 40 serial steps through LDS and one more read of the blocks.  Same
                // association as lds_dot<NX> (four partial sums per chunk).
                const double *nun = tabl + DC::CO + M::NTD + (size_t)(k + 1) * NX;
                constexpr int CH = NX <= 12 ? NX : (NX % 12 == 0 ? 12 : (NX % 11 == 0 ? 11 : (NX % 8 == 0 ? 8 : (NX % 7 == 0 ? 7 : 3))));
                double ac[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < NX; ++i) ac[(i % CH) & 3] = fma(dxc[i], nun[i], ac[(i % CH) & 3]);
                w[lay.rt + (size_t)k * NW + d] = (ac[0] + ac[1]) + (ac[2] + ac[3]);
            }
        }
        wave_sync();                                 // the next step's tables overwrite these
    }
}

// ---- the SQP loop of one instance, ONE WAVEFRONT, one launch: per round the linearisation at the current iterate (point pass on N
// lanes, direction pass on all of them), cost / residuals / stopping test, the QP by the Riccati interior-point method, the full step.
// (Until the middle of round 2 every round was a pair of launches and (max_iter + 1) of them were queued per solve: ~43 of the 51
// found nothing to do and cost 19 us of kernel boundaries each.)  The per-round scalars still travel through ws.state, which is
// what the phase functions read.
template <class M>
__global__ void __launch_bounds__(64, 1) chain_sqp_kernel(const LargeSpec sp, const LargeArgs a) {
    using Cfg = ChainCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NT = 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // Cfg::lds_doubles(N) doubles (launch_large)
    __shared__ int sidx[196];
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    ChainSolver<M> S(sp, lane);
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    S.bind_workspace(w, lay);
    S.setup(lds, sidx);
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = S.qmode ? a.u0fix + (size_t)inst * NU : nullptr;
    const int ne = (N + 1) * NW;
    const bool rti = (a.flags & 4) != 0;
    const int max_iter = rti ? 1 : sp.max_iter;
#ifdef MPCRL_PROFILE_PHASES
    __shared__ unsigned long long ph_buf[16];
    S.ph_init(ph_buf);
#endif
    S.ph0();
    int it, n_ipm, status;
    double cost, res[4];
    // opt-in divergence exit (mpcrl_set_exit_rule; the rule of small_solve_kernel and of the port): best residual so far, its value at
    // the last check, iterations to the next check
    double rbest = 1e300, rchk = 1e300;
    int exit_cnt = sp.exit_window;
    for (;;) {
        it = (int)S.state[ST_IT];
        n_ipm = (int)S.state[ST_NIPM];
        const bool last_tight = S.state[ST_TIGHT] != 0.0;
        const double stepn = S.state[ST_STEPN];
        // ---- linearisation at the current iterate, cost, NLP residuals
        wave_sync();
        typename ChainSolver<M>::RoundStart rs0;
        if constexpr (MPCRL_CHAIN_MERGE_CALLS != 0) {
            rs0 = ChainSolver<M>::round_call(S.ctx(), x0, u0f, sp.h, sp.rk_steps);
        } else {
            for (int e = lane; e < M::NTD; e += NT) lds[Cfg::oBig + DirCfg<M>::CO + e] = S.th[M::td_index(e)];   // (the QP phases reuse the region)
            if constexpr (Cfg::FUSE_GT)
                batched_pass<8>((N + 1) * NX, lane, [&](int e) { return S.NUv[e]; }, [&](int e, double v) { lds[Cfg::oBig + DirCfg<M>::CO + M::NTD + e] = v; });
            wave_sync();
            if (lane < N)
                chain_point_pass<M, false, true>(S.X, S.U, lds + Cfg::oBig + DirCfg<M>::CO, w, N, lane, sp.h, sp.rk_steps,
                                                 lds + Cfg::oBig + DirCfg<M>::CO + M::NTD + (Cfg::FUSE_GT ? (N + 1) * NX : 0));
            wave_sync();
            S.ph(6);
            chain_dir_pass<M>(S.th, w, lds + Cfg::oBig, N, lane, sp.h, sp.rk_steps);
            S.ph(8);
            rs0 = ChainSolver<M>::round_start_call(S.ctx(), x0, u0f);
        }
        cost = rs0.cost;
#pragma unroll
        for (int j = 0; j < 4; ++j) res[j] = rs0.res[j];
        S.ph(9);
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        status = -1;   // -1: carry on
        if (!(rmax < 1e300) || !(fabs(cost) < 1e300))   // the max-reductions drop NaNs, the cost sum does not
            status = 1;
        else if (rmax < sp.tol && last_tight && !(rti && it == 0))
            status = 0;
        else if (it >= max_iter)
            status = rmax < sp.tol ? 0 : 2;
        else if (sp.exit_window > 0) {
            rbest = fmin(rbest, rmax);
            if (it == 0)
                rchk = rmax;
            else if (--exit_cnt == 0) {
                if (rbest > sp.exit_factor * rchk) status = 2;   // no progress over the window: the SIMD is free for the next wavefront
                rchk = rbest, exit_cnt = sp.exit_window;
            }
        }
        if (status >= 0) break;
        // (MPCRL_EXACT_QP, test-only, wave-uniform: every QP to the tight tolerance from a cold interior-point start, fixed fraction to the boundary)
        const bool exact_qp = (a.flags & 64) != 0;
        const double rr_ = fmin(1.0, rmax), ad_ = (rmax < sp.tol || exact_qp) ? 0.0 : IPM_ADAPT_C * rr_ * rr_;
        const double tol_res = fmin(IPM_ADAPT_CAP, fmax(IPM_TOL_RES, ad_)), tol_mu = fmin(CHAIN_TOL_MU_FACTOR * IPM_ADAPT_CAP, fmax(IPM_TOL_MU, 1e-2 * ad_));
        const bool tight = tol_res <= IPM_TOL_RES && tol_mu <= IPM_TOL_MU;
        const double warm_mu = (stepn < 0.0 || exact_qp) ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
        S.frac_fixed = exact_qp;
        // the SQP Hessian: this lane's tiles of (R, Q) without c_k, in registers
        HessConst<M> hs;
        hs.th = S.th, hs.sck = S.sCK();
        // (MPCRL_COLD_DUAL keeps the stored multipliers of the dynamics while the QP starts from zero ones: not the same residual)
        const bool rg_ready = stepn >= 0.0 || !(a.flags & 16);
        if (!S.qp_solve(hs, x0, u0f, n_ipm, warm_mu, tol_res, tol_mu, rg_ready)) {
            status = 4;
            break;
        }
        double sl = 0.0;
        batched_pass<4>((N + 1) * NX, lane, [&](int e) { return Quad4{S.dx[e], S.X[e], S.nuq[e], 0.0}; },
                        [&](int e, const Quad4 &v) { sl = fmax(sl, fabs(v.a)), S.X[e] = v.b + v.a, S.NUv[e] = v.c; });
        batched_pass<2>(N * NU, lane, [&](int e) { return Pair2{S.du[e], S.U[e]}; },
                        [&](int e, const Pair2 &v) { sl = fmax(sl, fabs(v.a)), S.U[e] = v.b + v.a; });
        sl = wave_max(sl);
        if (lane == 0) S.state[ST_IT] = it + 1, S.state[ST_NIPM] = n_ipm, S.state[ST_TIGHT] = tight ? 1.0 : 0.0, S.state[ST_STEPN] = sl;
        S.ph(14);
    }
    // ---- finished (converged, failed or out of iterations): results + iterate
    double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    double *bnd = a.BND + (size_t)inst * 10 * nb;
    if (lane < NU) a.u0_out[(size_t)inst * NU + lane] = S.U[lane];
    if (lane == 0) {
        a.V[inst] = cost;
        a.status[inst] = status;
        if (a.iters) a.iters[inst * 2] = it, a.iters[inst * 2 + 1] = n_ipm;
        for (int j = 0; j < 4; ++j) a.RES[(size_t)inst * 4 + j] = res[j];
        S.state[ST_ACTIVE] = 0.0, S.state[ST_STATUS] = status;
    }
    // Lagrangian of the mirror, L = cost + pi' g + lam' h (nlp.py:1180; MPC.get_L, mpc.py:325-332): g_k = F(x_k, u_k) - x_{k+1}
    // is the r of this round's linearisation, h = -(slack of the bound row)
    double lag = 0.0;
    for (int e = lane; e < N * NX; e += NT) {
        const double pi_e = S.NUv[NX + e];
        PIg[e] = pi_e;
        lag = fma(pi_e, S.r[e], lag);
    }
    for (int e = lane; e < 2 * ne; e += NT) {
        const int sd = e / ne, ee = e - sd * ne, k = ee / NW, i = ee - k * NW;
        const bool h = !S.skipc(k, i) && S.has(sd, k, i);
        bnd[e] = h ? S.lam[e] : 0.0;
        bnd[2 * nb + e] = h ? S.t[e] : 1.0;
        if (h) lag = fma(-S.lam[e], S.bslack(sd, k, i, S.vc(k, i)), lag);
    }
    lag = wave_sum(lag);
    if (lane == 0 && a.LAG) a.LAG[inst] = cost + lag;
    for (int e = lane; e < 6 * ne; e += NT) bnd[4 * nb + e] = (e >= 4 * ne) ? 1.0 : 0.0;   // no soft rows here
    S.ph_flush();
}

}  // namespace mpcrl
