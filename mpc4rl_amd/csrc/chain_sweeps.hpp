// chain_sweeps.hpp — the QP of the chain solver: Mehrotra predictor-corrector, its Riccati sweeps as register-resident MFMA pipelines
// in Omega coordinates (OmCfg, chain_common.hpp), the bound rows in registers, and the phase calls of the SQP kernel.
#pragma once
#include "chain_common.hpp"

namespace mpcrl {

// Hessian source of the Riccati factorisation: the SQP uses c_k * (Q, R) from the parameter vector (an LDS table in the register
// layout); the sensitivities use the exact Lagrangian Hessian blocks chain_sens_ad_kernel wrote to the workspace (read one stage
// ahead).  Both hand out entries in Omega coordinates: register (rg, tj) of a lane holds entry (slot 4 rg + lane / 16,
// slot 16 tj + lane % 16); slots past NW (and the vector column) are zero.
template <class M>
struct HessConst {
    using O = OmCfg<M>;
    const double *th;
    const double *sck;
    double *tab;        // LDS, [RG * NT][64]: this lane's entries, unscaled (12 - 27 registers a lane would otherwise hold per sweep)
    int lane_;
    MPCRL_DI void begin(int lane) {
        const int lr = lane >> 4, lc = lane & 15;
        lane_ = lane;
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < O::NW && c < O::NW;
                tab[(rg * O::NT + tj) * 64 + lane] = in ? M::hess(false, O::nat(in ? e : 0), O::nat(in ? c : 0), th) : 0.0;
            }
    }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned) { th = S.th, sck = S.sCK(), tab = S.lds + ChainCfg<M>::oBig; }
    MPCRL_DI void prefetch(int) {}
    MPCRL_DI void advance(int) {}
    MPCRL_DI unsigned hex_offset() const { return 0u; }
    MPCRL_DI double tile(int k, int rg, int tj) const { return sck[k] * tab[(rg * O::NT + tj) * 64 + lane_]; }
    MPCRL_DI double term(int N, int i, int j) const { return sck[N] * M::Qs(th, i, j); }
};
template <class M>
struct HessGlobal {
    using O = OmCfg<M>;
    static constexpr int NW = O::NW, NU = O::NU;
    WsArr Hex;                      // [(N+1), NW, NW], stage-vector order [u; x]
    int lr, lc;
    double hn[O::RG][O::NT], hc[O::RG][O::NT];
    MPCRL_DI void begin(int lane) { lr = lane >> 4, lc = lane & 15; }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned hex_off) { Hex = S.arr(hex_off); }
    MPCRL_DI void prefetch(int k) {
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < NW && c < NW;
                const double v = Hex[k * NW * NW + (in ? O::nat(e) * NW + O::nat(c) : 0)];
                hn[rg][tj] = in ? v : 0.0;
            }
    }
    MPCRL_DI void advance(int k) {
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) hc[rg][tj] = hn[rg][tj];
        if (k > 0) prefetch(k - 1);
    }
    MPCRL_DI unsigned hex_offset() const { return Hex.off; }
    MPCRL_DI double tile(int, int rg, int tj) const { return hc[rg][tj]; }
    MPCRL_DI double term(int N, int i, int j) const { return Hex[N * NW * NW + (NU + i) * NW + NU + j]; }
};

template <class M>
struct ChainSolver {
    using Cfg = ChainCfg<M>;
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = 64, BAS = LargeLayout<M>::BAS;
    const LargeSpec *spp;   // kernel-argument copy of the problem (set-up only)
    const double *xs;       // x_ss (device)
    int N, lane;
    const double *th;   // full parameter vector of this instance
    bool qmode;
    bool frac_fixed = false;   // MPCRL_EXACT_QP (test-only): fixed fraction to the boundary
    // global (per instance)
    double *X, *U;
    WsArr NUv;             // NUv[k]: multiplier arriving at stage k, [(N+1)*NX] (index 0 unused)
    WsArr BA, r, q, dx, du, nuq, Dx, Du, rg, rb, rt, Dg, lam, t, aff, p, kff, state;
    WsArr G2, P2, hb2, minv2, mvu2;
    double *lds;
    // bounded rows: n0 / nm / ne coordinates carry a bound at stage 0 / 1..N-1 / N; coordinate lists in LDS (sidx)
    int n0, nm, ne, nrows;
    int *sidx;

    MPCRL_DI ChainSolver(const LargeSpec &s, int lane_) : spp(&s), xs(s.consts), N(s.N), lane(lane_) {}
    MPCRL_DI ChainSolver(const double *xs_, int N_, int lane_) : spp(nullptr), xs(xs_), N(N_), lane(lane_) {}   // inside a phase call: no set-up

#ifdef MPCRL_PROFILE_PHASES
    // per-wavefront tick counters in LDS (no global traffic inside the timed regions), flushed once by ph_flush()
    unsigned long long ph_t = 0;
    unsigned long long *ph_lds = nullptr;
    MPCRL_DI void ph0() { ph_t = clock64(); }
    MPCRL_DI void ph(int i) {
        const unsigned long long n_ = clock64();
        if (lane == 0) ph_lds[i] += n_ - ph_t;
        ph_t = n_;
    }
    MPCRL_DI void ph_init(unsigned long long *b) {
        ph_lds = b;
        if (lane < 16) b[lane] = 0;
        wave_sync();
    }
    MPCRL_DI void ph_flush() {
        wave_sync();
        if (lane < 16) atomicAdd(&g_phase_ticks[lane], ph_lds[lane]);
    }
#else
    MPCRL_DI void ph0() {}
    MPCRL_DI void ph(int) {}
    MPCRL_DI void ph_init(unsigned long long *) {}
    MPCRL_DI void ph_flush() {}
#endif

    MPCRL_DI double *sBA() const { return lds + Cfg::oBA; }
    MPCRL_DI double *sBB() const { return lds + Cfg::oBB; }
    MPCRL_DI double *sCK() const { return lds + Cfg::oCK; }
    MPCRL_DI double *sLB() const { return lds + Cfg::oLB; }          // lb, ub (stages 1..N-1), lbe, ube: NW each; lb0, ub0: 4 each

    MPCRL_DI double ck(int k) const { return sCK()[k]; }
    MPCRL_DI double ck_eval(int k) const {
        const LargeSpec &sp = *spp;
        if (sp.cost_kind == 0) return k == N ? 1.0 : sp.dT;                                            // nlp.py:1044-1055
        return k == 0 ? sp.dT : (k == N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);   // nlp.py:1083-1091
    }
    // bounds out of LDS (a lane-dependent index into kernel arguments would be copied to scratch)
    MPCRL_DI double lbv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sLB()[4 * NW + i] : -1e30;
        if (k == N) return i >= NU ? sLB()[2 * NW + i] : -1e30;
        return sLB()[i];
    }
    MPCRL_DI double ubv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sLB()[4 * NW + 4 + i] : 1e30;
        if (k == N) return i >= NU ? sLB()[3 * NW + i] : 1e30;
        return sLB()[NW + i];
    }
    MPCRL_DI bool has(int sd, int k, int i) const { return sd ? ubv(k, i) < NO_BOUND : lbv(k, i) > -NO_BOUND; }
    MPCRL_DI bool fixedc(int k, int i) const { return k == 0 && (i >= NU || qmode); }
    MPCRL_DI bool skipc(int k, int i) const { return k == N && i < NU; }
    MPCRL_DI double vc(int k, int i) const { return i < NU ? (k < N ? U[k * NU + i] : 0.0) : X[k * NX + i - NU]; }
    MPCRL_DI double dvc(const WsArr &ax, const WsArr &au, int k, int i) const {
        return i < NU ? (k < N ? au[k * NU + i] : 0.0) : ax[k * NX + i - NU];
    }
    MPCRL_DI double bslack(int sd, int k, int i, double v) const { return sd ? ubv(k, i) - v : v - lbv(k, i); }
    // row r of the compact list of bounded coordinates -> (stage, coordinate)
    MPCRL_DI void row_of(int r_, int &k, int &i) const {
        if (r_ < n0) {
            k = 0, i = sidx[r_];
        } else if (r_ < n0 + (N - 1) * nm) {
            const int q_ = r_ - n0;
            const int kk = q_ / nm;
            k = kk + 1, i = sidx[64 + q_ - kk * nm];
        } else
            k = N, i = sidx[128 + r_ - n0 - (N - 1) * nm];
    }
    MPCRL_DI double &LAM(int sd, int e) { return lam[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &TT(int sd, int e) { return t[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &AFF(int sd, int e) { return aff[sd * (N + 1) * NW + e]; }

    // 16-byte coalesced moves of one stage block: workspace -> registers -> LDS (n2 = number of 16-byte pieces)
    template <int NV>
    MPCRL_DI void blk_load(const WsArr src, int n2, d2_t (&rr)[NV], int lane) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int e2 = lane + 64 * s;
            (void)n2;
            rr[s] = *(const d2_t *)&src[2 * e2];   // past the block: the next array of the workspace (never used)
        }
    }
    template <int NV>
    MPCRL_DI void blk_to_lds(double *dst, int n2, const d2_t (&rr)[NV], int lane) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int e2 = lane + 64 * s;
            (void)n2;
            *(d2_t *)(dst + 2 * e2) = rr[s];       // the LDS regions leave room for 128 * NV doubles
        }
    }

    // ---- per-lane indices of the solver (a pure function of the lane and of the row counts the set-up left in LDS): what a phase
    // call re-derives on entry instead of receiving it
    MPCRL_DI void setup_lane(double *lds_, int *sidx_, bool read_counts) {
        lds = lds_, sidx = sidx_;
        if (read_counts) {
            n0 = (int)rfl((unsigned)sidx[192]), nm = (int)rfl((unsigned)sidx[193]), ne = (int)rfl((unsigned)sidx[194]);
            nrows = n0 + (N - 1) * nm + ne;
        }
    }
    // ---- one-off set-up: constants into LDS, GEMM tile of this lane, list of bounded coordinates
    MPCRL_DI void setup(double *lds_, int *sidx_) {
        const LargeSpec &sp = *spp;
        setup_lane(lds_, sidx_, false);
        if (lane <= N) sCK()[lane] = ck_eval(lane);
        if (lane == 0) {   // one lane, compile-time indices: the kernel arguments stay scalar operands
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                sLB()[i] = sp.lb[i], sLB()[NW + i] = sp.ub[i];
                sLB()[2 * NW + i] = i >= NU ? sp.lbe[i >= NU ? i - NU : 0] : -1e30, sLB()[3 * NW + i] = i >= NU ? sp.ube[i >= NU ? i - NU : 0] : 1e30;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sLB()[4 * NW + i] = sp.lb0[i], sLB()[4 * NW + 4 + i] = sp.ub0[i];
        }
        wave_sync();
        if (lane == 0) {
            int a = 0, b = 0, c = 0;
            for (int i = 0; i < NW; ++i) {
                if (i < NU && !qmode && (sLB()[4 * NW + (i < 4 ? i : 0)] > -NO_BOUND || sLB()[4 * NW + 4 + (i < 4 ? i : 0)] < NO_BOUND)) sidx[a++] = i;
                if (sLB()[i] > -NO_BOUND || sLB()[NW + i] < NO_BOUND) sidx[64 + b++] = i;
                if (i >= NU && (sLB()[2 * NW + i] > -NO_BOUND || sLB()[3 * NW + i] < NO_BOUND)) sidx[128 + c++] = i;
            }
            sidx[192] = a, sidx[193] = b, sidx[194] = c;
        }
        wave_sync();
        n0 = sidx[192], nm = sidx[193], ne = sidx[194];
        nrows = n0 + (N - 1) * nm + ne;
    }

    // ([B A]_k' nu_{k+1} - [0; nu_k])_i
    MPCRL_DI double GTnu(const WsArr &nu, int k, int i) const {
        double a = 0.0;
        if (k < N) {
            const WsArr Bk = BA + (k * BAS + i);
            for (int m = 0; m < NX; ++m) a = fma(Bk[m * NW], nu[(k + 1) * NX + m], a);
        }
        if (i >= NU && k > 0) a -= nu[k * NX + i - NU];
        return a;
    }

    // ---- start of an SQP round: q = c_k grad l_k, the cost, and the four NLP residual norms (stationarity, equality,
    // inequality, complementarity).  Q (symmetrised), X - x_ss and U of the whole horizon are staged in LDS; the stationarity
    // residual q + [B A]' nu_{k+1} - [0; nu_k] -+ lam is a stage-serial pass with [B A]_k staged through LDS (coalesced).
    MPCRL_DI double round_start(const double *x0, const double *u0f, double *res) {
        const int ne = (N + 1) * NW;
        double *lQ = lds + Cfg::oQ, *lX = lds + Cfg::oX, *lU = lds + Cfg::oU;
        for (int e = lane; e < NX * NX; e += NT) lQ[e] = M::Qs(th, e / NX, e % NX);
        batched_pass<8>((N + 1) * NX, lane, [&](int e) { return X[e]; }, [&](int e, double v) { lX[e] = v - xs[e % NX]; });
        batched_pass<2>(N * NU, lane, [&](int e) { return U[e]; }, [&](int e, double v) { lU[e] = v; });
        wave_sync();
        double val = 0.0;
        batched_pass<4>(ne, lane, [&](int e) { return Pair2{lam[e], lam[ne + e]}; }, [&](int e, const Pair2 &lm) {
            const int k = e / NW, i = e - k * NW;
            const bool term = k == N;
            double a = 0.0, v = 0.0;
            if (i < NU) {
                if (!term) {
#pragma unroll
                    for (int j = 0; j < NU; ++j) a = fma(M::Rs(th, i, j), lU[k * NU + j], a);
                    v = lU[k * NU + i];
                }
            } else {
                const double *qr = lQ + (i - NU) * NX, *xk = lX + k * NX;
                a = lds_dot<NX>(qr, 1, xk, 0.0);
                v = xk[i - NU];
            }
            const double qe = ck(k) * a;
            q[e] = qe;
            val = fma(0.5 * qe, v, val);
            double g = qe;
            if (!skipc(k, i)) {
                if (has(0, k, i)) g -= lm.a;
                if (has(1, k, i)) g += lm.b;
            }
            rg[e] = g;   // q -+ lam: the stage pass below adds the multiplier terms of the dynamics
        });
        wave_sync();
        double rs = 0, re = 0, ri = 0, rc = 0;
        if constexpr (Cfg::FUSE_GT) {
            // [B A]_k' nu_{k+1} arrives in rt from the direction pass of this round (chain_dir_pass): one pass over the entries
            batched_pass<4>(ne, lane,
                            [&](int e) {
                                const int k = e / NW, i = e - k * NW;
                                return Quad4{rg[e], (i >= NU && k > 0) ? NUv[k * NX + i - NU] : 0.0, k < N ? rt[e] : 0.0, 0.0};
                            },
                            [&](int e, const Quad4 &v) {
                                const int k = e / NW, i = e - k * NW;
                                if (k == N && i < NU) return;           // (no controls at the terminal stage)
                                const double a = (v.a - v.b) + v.c;
                                const bool fx = k < N && fixedc(k, i);
                                if (!fx) rs = fmax(rs, fabs(a));
                                // the vector itself stays in rg: it IS the stationarity residual the next QP starts from (qp_start_residuals)
                                rg[e] = fx ? 0.0 : a;
                            });
        } else {
            d2_t nB[Cfg::DEPTH][Cfg::NBA2];
            double ng[Cfg::DEPTH], nn[Cfg::DEPTH], no[Cfg::DEPTH];
            double *lBA = sBA(), *lnu = sBB();
            const int lj = lane < NW ? lane : 0, lx = lane < NX ? lane : 0;
            staged_loop<Cfg::DEPTH>(
                N,
                [&](int k, auto sl) {
                    constexpr int d = decltype(sl)::value;
                    blk_load(BA + k * BAS, NX * NW / 2, nB[d], lane);
                    ng[d] = rg[k * NW + lj], nn[d] = NUv[(k + 1) * NX + lx];
                    {
                        const bool c_ = lane >= NU && lane < NW && k > 0;
                        const double t_ = NUv[c_ ? k * NX + lj - NU : 0];
                        no[d] = c_ ? t_ : 0.0;
                    }
                },
                [&](int k, auto sl, auto refill) {
                    constexpr int d = decltype(sl)::value;
                    blk_to_lds(lBA, NX * NW / 2, nB[d], lane);
                    if (lane < NX) lnu[lane] = nn[d];
                    double a = ng[d] - no[d];
                    refill();
                    wave_sync();
                    a = lds_dot<NX>(lBA + lj, NW, lnu, a);
                    if (lane < NW && !fixedc(k, lane)) rs = fmax(rs, fabs(a));
                    // the vector itself stays in rg: it IS the stationarity residual the next QP starts from (qp_start_residuals)
                    if (lane < NW) rg[k * NW + lane] = fixedc(k, lane) ? 0.0 : a;
                    wave_sync();
                });
            if (lane >= NU && lane < NW) {
                const double a = rg[N * NW + lane] - NUv[N * NX + lane - NU];
                rs = fmax(rs, fabs(a));
                rg[N * NW + lane] = a;
            }
        }
        for (int r_ = lane; r_ < nrows; r_ += NT) {
            int k, i;
            row_of(r_, k, i);
            const int e = k * NW + i;
            const double v = vc(k, i);
            if (has(0, k, i)) {
                const double h = lbv(k, i) - v;
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[e] * h));
            }
            if (has(1, k, i)) {
                const double h = v - ubv(k, i);
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[ne + e] * h));
            }
        }
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int, double v) { re = fmax(re, fabs(v)); });
        if (lane < NX) re = fmax(re, fabs(X[lane] - x0[lane]));
        if (qmode && lane < NU) re = fmax(re, fabs(U[lane] - u0f[lane]));
        res[0] = wave_max(rs), res[1] = wave_max(re), res[2] = wave_max(ri), res[3] = wave_max(rc);
        return wave_sum(val);
    }

    // =====================================================================================================================
    // The Riccati sweeps: register-resident MFMA pipelines in Omega coordinates (OmCfg).  Workspace in, workspace out (natural-order
    // p, kff, Dx, Du for the row phases); no LDS inside the stage loops — a lone wavefront has nobody to hide an LDS round trip
    // behind (round 3 staged its operands through LDS: 69 s_waitcnt per factor stage for 49 MFMAs).
    // =====================================================================================================================
    typedef double d4_t __attribute__((ext_vector_type(4)));
    template <int SRC>
    MPCRL_DI static double bcast_lane(double v) {   // the value lane SRC holds, in every lane (two v_readlane_b32)
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, SRC), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), SRC);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    // state index of row slot 4 rg + lr of a register (pad rows of the control group: some valid address, the value is masked)
    template <int RG_>
    MPCRL_DI static int om_xr(int lr) { return 4 * RG_ + lr - (RG_ >= OmCfg<M>::GQ ? NU : 0); }
    // stage-vector index [u; x] of slot 4 rg + lr
    template <int RG_>
    MPCRL_DI static int om_nat(int lr) {
        using O = OmCfg<M>;
        if constexpr (RG_ < O::GQ) return NU + 4 * RG_ + lr;
        if constexpr (RG_ > O::GQ) return 4 * RG_ + lr;
        return lr < NU ? lr : O::Q + lr;
    }

    // ---- factor sweep (see OmCfg): P_{k+1} stays in registers in the result layout, which is its operand layout for T = P W.
    // slot of entry i of the stage vector [u; x]
    MPCRL_DI static int om_slot(int i) { return i < NU ? OmCfg<M>::Q + i : (i - NU < OmCfg<M>::Q ? i - NU : i); }

    // ---- y_k = [B A]_k' nu_{k+1}, k = 0 .. N - 1, for a multiplier array nu [(N+1) NX]: into ly (LDS, [k HBS + slot of the stage
    // vector]); lnu (LDS) receives nu in Omega order.  Stage-parallel (no chain): [B A]_k as it lies in the workspace is the A
    // operand, the vector the B operand, 4 stages of operands in flight.
    MPCRL_DI void wt_nu_pass(const WsArr nu, double *lnu, double *ly) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        stage_vec_lds<true>(lnu, nu, N);
        wave_sync();
        int colnat[NTR];
#pragma unroll
        for (int tj = 0; tj < NTR; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const int rbase = lr * NW;
        double nA[D][RG][NTR];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) nA[d][rg][tj] = BA[k * BAS + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW];
                });
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                double Ak[RG][NTR], vop[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) {
                        double v = nA[d][rg][tj];
                        if constexpr (rg == GQ) v = padl ? 0.0 : v;
                        if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                        Ak[rg][tj] = v;
                    }
                    vop[rg] = lnu[(k + 1) * O::HBS + 4 * rg + lr];
                });
                refill();
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[ks][ti], vop[ks], acc[ti], 0, 0, 0);
                if (lc == 0)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        ly[k * O::HBS + 4 * rg + lr] = acc[rg / 4][rg % 4];
                    });
            });
        wave_sync();
    }

    // ---- residuals of the QP at its START (what qp_solve asks for with the residual scaling on: later iterations scale them):
    // there the direction of every stage but the first is zero — dx_0 = x0 - X_0 and a pinned du_0 are the only entries of (dx, du)
    // the QP starts with — so  rb_k = r_k,  rg_k = q_k -+ lam + [B A]_k' nuq_{k+1} - [0; nuq_k], plus [B A]_0 dv_0 and c_0 H dv_0 at
    // stage 0 where dv_0 != 0 (first QP of a warm solve from a new initial state).  One MFMA pass + one pass over the entries,
    // against a stage-serial pass with [B A]_k staged through LDS (round 3: 7 % / 14 % of the kernel at n_mass 5 / 7).
    MPCRL_DI double qp_residuals2() {
        using O = OmCfg<M>;
        const int ne = (N + 1) * NW;
        double *const lnu = lds + Cfg::oBig, *const ly = lnu + (N + 1) * O::HBS;
        wt_nu_pass(nuq, lnu, ly);
        double rloc = 0.0;
        batched_pass<4>(ne, lane,
                        [&](int e) {
                            const int k = e / NW, i = e - k * NW;
                            return Quad4{q[e], lam[e], lam[ne + e], (i >= NU && k > 0) ? nuq[k * NX + i - NU] : 0.0};
                        },
                        [&](int e, const Quad4 &v) {
                            const int k = e / NW, i = e - k * NW;
                            double g = v.a;
                            if (!skipc(k, i)) {
                                if (has(0, k, i)) g -= v.b;
                                if (has(1, k, i)) g += v.c;
                            }
                            g += (k < N ? ly[k * O::HBS + om_slot(i)] : 0.0) - v.d;
                            if (fixedc(k, i) || skipc(k, i)) g = 0.0;
                            rg[e] = g, rloc = fmax(rloc, fabs(g));
                        });
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int e, double v) { rb[e] = v, rloc = fmax(rloc, fabs(v)); });
        wave_sync();
        rloc = fmax(rloc, stage0_direction_terms());
        return rloc;
    }

    // ---- the same at no pass over the [B A]_k: round_start evaluated rg at the multipliers the QP starts from (qp_solve, rg_ready)
    MPCRL_DI double qp_start_residuals() {
        const int ne = (N + 1) * NW;
        double rloc = 0.0;
        batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int, double v) { rloc = fmax(rloc, fabs(v)); });
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int e, double v) { rb[e] = v, rloc = fmax(rloc, fabs(v)); });
        wave_sync();
        return fmax(rloc, stage0_direction_terms());
    }
    // stage 0 with a non-zero direction (dx_0 = x0 - X_0 of a warm solve from a new state, a pinned du_0): [B A]_0 dv_0 into rb_0,
    // c_0 H dv_0 into rg_0; one row / column per lane.  Returns the largest entry it changed.
    MPCRL_DI double stage0_direction_terms() {
        double rloc = 0.0, d0 = 0.0;
        if (lane < NX) d0 = fabs(dx[lane]);
        if (lane < NU) d0 = fmax(d0, fabs(du[lane]));
        if (wave_max(d0) > 0.0) {
            double dv[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) dv[j] = j < NU ? du[j < NU ? j : 0] : dx[j >= NU ? j - NU : 0];
            if (lane < NX) {
                double a = rb[lane];
#pragma unroll 4
                for (int j = 0; j < NW; ++j) a = fma(BA[lane * NW + j], dv[j], a);
                rb[lane] = a, rloc = fmax(rloc, fabs(a));
            }
            if (lane < NW && !fixedc(0, lane)) {
                double hd = 0.0;
#pragma unroll 4
                for (int j = 0; j < NW; ++j) hd = fma(M::hess(false, lane < NW ? lane : 0, j, th), dv[j], hd);
                const double g = fma(ck(0), hd, rg[lane]);
                rg[lane] = g, rloc = fmax(rloc, fabs(g));
            }
            wave_sync();
        }
        return rloc;
    }

    // SENS = false (the QPs of the SQP): the sweep leaves -K_k in the rows NX.. of the stage block of [B A]_k (LargeLayout::BAS) — the
    // vector sweeps multiply with [A B] and K; nothing of the size of a stage block is written.  SENS = true (the adjoint
    // factorisation of the sensitivities): the closed-loop block G_k and P_k go to HBM for forward2_sens.
    template <class HS, bool SENS>
    MPCRL_DI bool factor2(HS &hs, const WsArr g, const WsArr bb) {
        constexpr bool STORE_P = SENS;
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NT = O::NT, NTR = O::NTR, GQ = O::GQ, TQ = O::TQ, RQ = O::RQ, LQ = O::LQ, TV = O::TV, LV = O::LV;
        constexpr int FD = NTR <= 2 ? 2 : 1;     // prefetch slots (a stage is 3 - 9 us of work: one stage ahead covers the HBM latency; registers at n_mass 7)
        constexpr bool RAGGED = 4 * RG > NW;     // the last row group runs past NW
        const int lr = lane >> 4, lc = lane & 15;
        hs.begin(lane);
        bool ok = true;
        int colnat[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const bool vcl = lc == LV, wcl = lc < LV, padl = lr < NU, ucl = lc >= LQ && lc < LQ + NU;
        const double vcm = vcl ? 1.0 : 0.0;
        const int rbase = lr * NW;
        int btoff[NTR];
        bool btok[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int e = 16 * ti + lc, xr = O::xrow(e < NW ? e : 0);
            btok[ti] = e < NW && xr >= 0 && lr < NU;
            btoff[ti] = (btok[ti] ? xr : 0) * NW + (lr < NU ? lr : 0);
        }
        const unsigned gbase = (unsigned)lane, cbase = (unsigned)(lr * LV + (lc < LV ? lc : 0));
        // ---- terminal stage: P_N = c_N hess l_N + D_N, p_N = g_N
        d4_t Pt[NTR][NT];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    double v = 0.0;
                    if constexpr (rg < RG) {
                        const int e = 4 * rg + lr, c = 16 * tj + lc;
                        const int xr = O::xrow(e < NW ? e : 0), xc = O::xrow(c < NW ? c : 0);
                        const bool rok = e < NW && xr >= 0;
                        if (rok && c < NW && xc >= 0) {
                            v = hs.term(N, xr > xc ? xr : xc, xr > xc ? xc : xr);
                            if (xr == xc) v += Dg[N * NW + NU + xr];
                        } else if (rok && c == O::VC)
                            v = g[N * NW + NU + xr];
                        if (STORE_P && (tj < TV || lc < LV)) P2[N * O::GSZ + O::goff(rg, tj, lr, lc)] = v;
                        if (rok && c == O::VC) p[N * NX + xr] = v;
                    }
                    Pt[ti][tj][r] = v;
                }
            });
        });
        // per stage and lane: W (RG x NT registers; in the tile of the vector column the lane of that column fetches b instead),
        // B' (NTR), and ONE register per row group for the right-hand side and the barrier diagonal (the lane of the vector column
        // fetches g, the diagonal lane D: they are different lanes for every valid row)
        // (one lane is both: the diagonal lane of row 16 ti + LV in a tile ti != TV is the lane of the vector column — its D comes
        // with a load of its own, ndx)
        double nW[FD][RG][NT], nBt[FD][NTR], ngd[FD][RG], ndx[FD][NTR];
        const unsigned bbrel = bb.off - BA.off, grel = g.off - BA.off, dgrel = Dg.off - BA.off;
        hs.prefetch(N - 1);
        staged_loop<FD>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const WsArr Bk = BA + k * BAS;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const int xr = om_xr<rg>(lr), nt = om_nat<rg>(lr);
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        const int ow = k * BAS + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW;
                        nW[d][rg][tj] = BA[(tj == TV && vcl) ? (int)bbrel + k * NX + xr : ow];
                    }
                    const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                    ngd[d][rg] = BA[(int)(vcl ? grel : dgrel) + k * NW + (rok ? nt : 0)];
                });
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) {
                    if constexpr (SENS) nBt[d][ti] = Bk[btoff[ti]];
                    ndx[d][ti] = (ti != TV && 16 * ti + LV < NW) ? Dg[k * NW + O::nat(16 * ti + LV < NW ? 16 * ti + LV : 0)] : 0.0;
                }
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const bool pin = k == 0 && qmode;
                hs.advance(k);
                // ---- the stage operands out of their prefetch slot: W = [A B | b] with its pad rows, B' as an A operand
                d4_t Wt[NTR][NT];
                double gd[RG], Bt[NTR], dgx[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) dgx[ti] = ndx[d][ti];
                static_for<NTR>([&](auto ti_) {
                    static_for<4>([&](auto r_) {
                        constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) {
                            double v = 0.0;
                            if constexpr (rg < RG) {
                                v = nW[d][rg][tj];
                                if (tj == TV) v = lc <= LV ? v : 0.0;
                                if constexpr (rg == GQ) v = padl ? 0.0 : v;
                                if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                            }
                            Wt[ti][tj][r] = v;
                        }
                        if constexpr (rg < RG) {
                            const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                            gd[rg] = rok ? ngd[d][rg] : 0.0;
                        }
                    });
                });
                if constexpr (SENS) {
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) Bt[ti] = btok[ti] ? nBt[d][ti] : 0.0;
                }
                refill();
                // ---- T = P W, row tile by row tile (the column tile of P it read is dead afterwards)
                d4_t Tt[NTR][NT];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[ks / 4][ti][ks % 4], Wt[ks / 4][tj][ks % 4], acc, 0, 0, 0);
                        Tt[ti][tj] = acc;
                    }
                ph(10);
                // its vector column is P b (kept: hb, stored below), then + p
                double hbv[RG];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * ti + r < RG) {
                            hbv[4 * ti + r] = Tt[ti][TV][r];
                            Tt[ti][TV][r] = fma(Pt[ti][TV][r], vcm, Tt[ti][TV][r]);
                        }
                // ---- M = H + D + W' T, vector column g + W' (P b + p); column tile by column tile (T's is dead afterwards).  The column
                // tile of the control group goes first: its diagonal tile feeds the Cholesky below, whose serial VALU chain (~1.5 k cycles)
                // then runs in the shadow of the other column tiles' MFMAs
                d4_t Mt[NTR][NT];
                auto m_column = [&](auto tj_) {
                    constexpr int tj = decltype(tj_)::value;
                    static_for<NTR>([&](auto tmo_) {
                        constexpr int tmo = decltype(tmo_)::value, tm = tmo == 0 ? TQ : (tmo <= TQ ? tmo - 1 : tmo);     // row tile TQ first
                        d4_t acc;
                        static_for<4>([&](auto r_) {
                            constexpr int r = decltype(r_)::value, rg = 4 * tm + r;
                            double c = 0.0;
                            if constexpr (rg < RG) {
                                c = hs.tile(k, rg, tj);
                                if (tj == tm) c += (lc == 4 * r + lr) ? ((tm != TV && r == LV / 4 && vcl) ? dgx[tm] : gd[rg]) : 0.0;
                                if (tj == TV) c += vcl ? gd[rg] : 0.0;
                            }
                            acc[r] = c;
                        });
#pragma unroll
                        for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[ks / 4][tm][ks % 4], Tt[ks / 4][tj][ks % 4], acc, 0, 0, 0);
                        Mt[tm][tj] = acc;
                    });
                };
                m_column(std::integral_constant<int, TQ>{});
                // ---- Cholesky of the control block on broadcast values (every lane, redundantly), then its inverse
                double Minv[NU][NU];
                {
                    const double src = Mt[TQ][TQ][RQ];
                    double a_[NU][NU], Lc[NU][NU], Li[NU][NU];
                    static_for<NU>([&](auto i_) {
                        static_for<NU>([&](auto j_) {
                            constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
                            if constexpr (j <= i) a_[i][j] = bcast_lane<16 * i + LQ + j>(src);
                        });
                    });
                    bool okc = true;
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double a = a_[i][j];
#pragma unroll
                            for (int m = 0; m < j; ++m) a -= Lc[i][m] * Lc[j][m];
                            if (i == j) {
                                okc = okc && (a > 0.0);
                                // 1 / sqrt(a) from the hardware seed + two Newton steps (~1 ulp): the pivot is a positive normal number
                                // wherever the result is used, and an IEEE sqrt followed by an IEEE division is ~40 dependent
                                // instructions on the critical path of every stage
                                double y = __builtin_amdgcn_rsq(a);
                                y = y * fma(-0.5 * a * y, y, 1.5);
                                Lc[i][i] = y * fma(-0.5 * a * y, y, 1.5);
                            } else
                                Lc[i][j] = a * Lc[j][j];
                        }
                    ok = ok && (okc || pin);
                    // Li = L^-1 (lower; Lc carries the inverted diagonal), Minv = Li' Li
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        Li[j][j] = Lc[j][j];
#pragma unroll
                        for (int i = j + 1; i < NU; ++i) {
                            double a = 0.0;
#pragma unroll
                            for (int m = j; m < i; ++m) a -= Lc[i][m] * Li[m][j];
                            Li[i][j] = a * Lc[i][i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double a = 0.0;
#pragma unroll
                            for (int m = i; m < NU; ++m) a = fma(Li[m][i], Li[m][j], a);
                            Minv[i][j] = a, Minv[j][i] = a;
                        }
                }
                static_for<NT>([&](auto tj_) {
                    if constexpr (decltype(tj_)::value != TQ) m_column(tj_);
                });
                ph(11);
                double minvop = 0.0;     // A operand of K = R^-1 [S | R | mv_u]: R^-1(i, l) at lane (lr = l, lc = i)
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int l = 0; l < NU; ++l) minvop = (lc == i && lr == l) ? Minv[i][l] : minvop;
                if (pin) minvop = 0.0;
                ph(12);
                // ---- K (register 0 of one MFMA per column tile), the rank-NU update P' = M - S' K, G = W - B K with -K in the pad rows
                double nK[NT], nKz[NT], Sr[NTR];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const d4_t z4 = {0.0, 0.0, 0.0, 0.0};
                    const d4_t kt = __builtin_amdgcn_mfma_f64_16x16x4f64(minvop, Mt[TQ][tj][RQ], z4, 0, 0, 0);
                    nK[tj] = -kt[0];
                    nKz[tj] = (tj == TQ && ucl) ? 0.0 : nK[tj];
                }
#pragma unroll
                for (int ta = 0; ta < NTR; ++ta) Sr[ta] = Mt[TQ][ta][RQ];
#pragma unroll
                for (int ta = 0; ta < NTR; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb) Mt[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Sr[ta], nK[tb], Mt[ta][tb], 0, 0, 0);
                if constexpr (SENS) {
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) Wt[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bt[ti], nKz[tj], Wt[ti][tj], 0, 0, 0);
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) Wt[TQ][tj][RQ] = padl ? nKz[tj] : Wt[TQ][tj][RQ];
                    // ---- out: G_k and P_k in the register layout (full 512-byte bursts)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
#pragma unroll
                        for (int tj = 0; tj < TV; ++tj) {
                            G2[k * O::GSZ + (rg * TV + tj) * 64 + gbase] = Wt[rg / 4][tj][rg % 4];
                            P2[k * O::GSZ + (rg * TV + tj) * 64 + gbase] = Mt[rg / 4][tj][rg % 4];
                        }
                    });
                    if (wcl)      // the last column tile: its columns < NW, compactly
                        static_for<RG>([&](auto rg_) {
                            constexpr int rg = decltype(rg_)::value;
                            G2[k * O::GSZ + RG * TV * 64 + rg * O::CT + cbase] = Wt[rg / 4][TV][rg % 4];
                            P2[k * O::GSZ + RG * TV * 64 + rg * O::CT + cbase] = Mt[rg / 4][TV][rg % 4];
                        });
                } else {
                    // ---- out: -K_k into rows NX .. NX + NU - 1 of the stage block, natural column order, zeros in the control columns
                    // (lane (lr = l, lc) holds -K(l, slot 16 tj + lc); columns past NW and the vector column go to the dump slot)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        const bool cok = padl && 16 * tj + lc < NW;
                        BA[cok ? k * BAS + (NX + lr) * NW + colnat[tj] : N * BAS] = nKz[tj];
                    }
                }
                // from the lanes of the vector column hb_k, and p_k, kff_k in natural order for the other phases; R^-1 from the lanes
                // that hold its entries
                if (vcl) {
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        hb2[k * O::HBS + 4 * rg + lr] = hbv[rg];
                        // natural p: the pad rows (and rows past NW) go to a dump slot behind the array of stage N
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        p[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = Mt[rg / 4][TV][rg % 4];
                    });
                    kff[padl ? k * NU + lr : N * NU] = -nK[TV];
                }
                if (lc < NU && lr < NU) minv2[k * 16 + 4 * lc + lr] = minvop;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) Pt[ti][tj] = Mt[ti][tj];
                ph(13);
            });
        return ok;
    }

    // One vector of the whole horizon from the workspace into LDS in Omega order, dst[k HBS + slot]: natural stage-vector arrays
    // (stride NW: slot -> nat(slot)) or state arrays (stride NX: slot -> state index, control slots 0); kmax = last stage.
    template <bool STATE>
    MPCRL_DI void stage_vec_lds(double *dst, const WsArr src, int kmax) {
        using O = OmCfg<M>;
        const int n = (kmax + 1) * O::HBS;
        batched_pass<4>(n, lane,
                        [&](int e) {
                            const int k = e / O::HBS, sl = e - k * O::HBS;
                            if (STATE) {
                                const int xr = O::xrow(sl < NW ? sl : 0);
                                const double v = src[k * NX + (sl < NW && xr >= 0 ? xr : 0)];
                                return (sl < NW && xr >= 0) ? v : 0.0;
                            } else {
                                const double v = src[k * NW + (sl < NW ? O::nat(sl) : 0)];
                                return sl < NW ? v : 0.0;
                            }
                        },
                        [&](int e, double v) { dst[e] = v; });
    }

    // ---- backward vector sweep for a new right-hand side g on the factors of the current iteration (K_k in the stage block, hb_k):
    //     [A'v; mv_u] = g + [A B]_k' v,  v = p_{k+1} + hb_k          p_k = (g_x + A'v) - K_k' mv_u          kff_k = R_k^-1 mv_u
    // (= g_x - K'g_u + Acl'v with Acl = A - B K, which is never formed).  Two dependent groups of MFMAs per stage: [A B]' v with
    // [A B]_k as it lies in the workspace as the A operand and the previous result registers as the B operand (column 0 of the lanes
    // carries the vector; the 16 columns are identical copies), then the rank-NU term with the lanes (lr = l, lc) holding -K(l, .) as
    // the A operand and the control rows of the first result as the B operand.  The two vectors of the horizon (g, hb) are staged in
    // LDS in Omega order first: the stream of the stage blocks is all that is left in the global-memory queue.
    MPCRL_DI void backward_vec2(const WsArr g) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;   // (stages in flight: registers at n_mass 7)
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        double *const lg = lds + Cfg::oBig, *const lhb = lg + (N + 1) * O::HBS;
        stage_vec_lds<false>(lg, g, N);
        batched_pass<4>(N * O::HBS, lane, [&](int e) { return hb2[e]; }, [&](int e, double v) { lhb[e] = v; });
        wave_sync();
        int colnat[NTR];
#pragma unroll
        for (int tj = 0; tj < NTR; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const int rbase = lr * NW, kbase = (NX + (padl ? lr : 0)) * NW;
        d4_t R[NTR];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                double v = 0.0;
                if constexpr (rg < RG) {
                    v = lg[N * O::HBS + 4 * rg + lr];
                    if constexpr (rg == GQ) v = padl ? 0.0 : v;
                }
                R[ti][r] = v;
            });
        });
        if (lc == 0)
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                p[rok ? N * NX + om_xr<rg>(lr) : (N + 1) * NX] = R[rg / 4][rg % 4];
            });
        double nA[D][RG][NTR], nK[D][NTR];
        staged_loop<D>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) nA[d][rg][tj] = BA[k * BAS + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW];
                });
#pragma unroll
                for (int tj = 0; tj < NTR; ++tj) nK[d][tj] = BA[k * BAS + kbase + colnat[tj]];
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                double Ak[RG][NTR], Kt[NTR], vop[RG], gt[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) {
                        double v = nA[d][rg][tj];
                        if constexpr (rg == GQ) v = padl ? 0.0 : v;                                  // pad rows of W
                        if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                        if (RAGGED) v = 16 * tj + lc < NW ? v : 0.0;                                 // (columns past NW: result rows that feed the next operand)
                        Ak[rg][tj] = v;
                    }
                    gt[rg] = lg[k * O::HBS + 4 * rg + lr];
                    double v = R[rg / 4][rg % 4] + lhb[k * O::HBS + 4 * rg + lr];
                    if constexpr (rg == GQ) v = padl ? 0.0 : v;
                    vop[rg] = v;
                });
#pragma unroll
                for (int tj = 0; tj < NTR; ++tj) {
                    const double v = nK[d][tj];
                    Kt[tj] = (padl && 16 * tj + lc < NW) ? v : 0.0;
                }
                refill();
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ti][r] = 4 * ti + r < RG ? gt[4 * ti + r < RG ? 4 * ti + r : 0] : 0.0;
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[ks][ti], vop[ks], acc[ti], 0, 0, 0);
                // the control rows now hold mv_u = g_u + B'v (lanes lr < NU of the group's register): operand of the rank-NU term
                const double mvu = acc[GQ / 4][GQ % 4];
                const double mop = padl ? mvu : 0.0;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kt[ti], mop, acc[ti], 0, 0, 0);
                if (lc == 0)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        p[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = v;
                        if constexpr (rg == GQ) mvu2[k * 4 + lr] = v;      // (lane lr = 3 of the group: a state row, never read)
                    });
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) R[ti] = acc[ti];
            });
        wave_sync();
        // feed-forward kff_k = R_k^-1 mv_u, one stage per lane
        for (int k = lane; k < N; k += NT) {
            double mv[NU];
#pragma unroll
            for (int m = 0; m < NU; ++m) mv[m] = mvu2[k * 4 + m];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < NU; ++m) a = fma(minv2[k * 16 + 4 * i + m], mv[m], a);
                kff[k * NU + i] = a;     // (R^-1 is stored as zero at a pinned stage 0)
            }
        }
        wave_sync();
    }

    // ---- forward sweep on the same blocks:  du_k = -kff_k - K_k dx_k,   dx_{k+1} = b_k + A_k dx_k + B_k du_k.
    // One A operand per (row tile, k-step) serves both: G0 = [A B; -K 0] in Omega coordinates (x rows from [B A]_k, control rows from
    // the K rows of the stage block: one gather, the lane's row decides which), contraction over its COLUMNS — the transposed access
    // pattern of the block.  Phase 1: [A dx + b; du] = [b; -kff] + G0 [dx; 0]; phase 2 adds B du: the control k-step again with du
    // in the control slots of the operand.
    MPCRL_DI void forward2(const WsArr bb) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        // staged in LDS, Omega order: [b_k; kff_k] (control slots: kff)
        double *const lbk = lds + Cfg::oBig;
        stage_vec_lds<true>(lbk, bb, N - 1);
        wave_sync();
        for (int e = lane; e < N * NU; e += NT) {
            const int k = e / NU;
            lbk[k * O::HBS + O::Q + (e - k * NU)] = kff[e];
        }
        wave_sync();
        // element (row slot a, column slot b) of G0: this lane wants a = 16 ti + lc, b = 4 ks + lr
        int rowoff[NTR], coloff[RG];
        bool rowok[NTR], colok[RG];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int a = 16 * ti + lc, ac = a < NW ? a : 0, xr = O::xrow(ac);
            rowok[ti] = a < NW;
            rowoff[ti] = (xr >= 0 ? xr : NX + (ac - O::Q)) * NW;      // a state row of [B A], or row NX + l of the K rows
        }
#pragma unroll
        for (int ks = 0; ks < RG; ++ks) {
            const int b_ = 4 * ks + lr;
            colok[ks] = b_ < NW;
            coloff[ks] = O::nat(b_ < NW ? b_ : 0);
        }
        if (lane < NX) Dx[lane] = 0.0;
        double w[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) w[rg] = 0.0;
        double nGt[D][NTR][RG];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks) nGt[d][ti][ks] = BA[k * BAS + rowoff[ti] + coloff[ks]];
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                d4_t acc[NTR];
                static_for<NTR>([&](auto ti_) {
                    static_for<4>([&](auto r_) {
                        constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                        double c = 0.0;
                        if constexpr (rg < RG) {
                            c = lbk[k * O::HBS + 4 * rg + lr];
                            if constexpr (rg == GQ) c = padl ? -c : c;       // -kff in the control slots
                        }
                        acc[ti][r] = c;
                    });
                });
                double Gk[NTR][RG];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks) {
                        const double v = nGt[d][ti][ks];
                        Gk[ti][ks] = (!RAGGED || (rowok[ti] && colok[ks])) ? v : 0.0;
                    }
                refill();
                w[GQ] = padl ? 0.0 : w[GQ];                                   // phase 1: the control slots of the operand are empty
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ti][ks], w[ks], acc[ti], 0, 0, 0);
                const double du_ = acc[GQ / 4][GQ % 4];                       // control rows: du_k = -kff_k - K_k dx_k
                const double w2 = padl ? du_ : 0.0;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ti][GQ], w2, acc[ti], 0, 0, 0);   // + B du
                if (lc == 0)      // the vector sits in column 0: rows past NW go to the dump slot
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        if constexpr (rg == GQ)
                            Dx[padl ? (int)(Du.off - Dx.off) + k * NU + lr : (k + 1) * NX + om_xr<rg>(lr)] = v;     // du_k / dx_{k+1}
                        else
                            Dx[rok ? (k + 1) * NX + om_xr<rg>(lr) : (N + 1) * NX] = v;
                    });
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) w[rg] = acc[rg / 4][rg % 4];
            });
        wave_sync();
    }

    // ---- the NU adjoint solves of the sensitivities in ONE forward sweep: right-hand side -e_iu in the controls of stage 0, no
    // dynamics offset, so p_k = 0 and kff_k = 0 for k >= 1 and kff_0 = -R_0^-1 e_iu; solve iu rides in column iu of the B operand
    // (the 16 columns of the MFMA cost the same as one).  Out: Ydx / Ydu / Ydnu [iu][...] as chain_sens_mix / chain_sens_out read them.
    MPCRL_DI void forward2_sens(const WsArr Ydx, const WsArr Ydu, const WsArr Ydnu) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, TV = O::TV, LV = O::LV, D = NTR <= 2 ? 3 : 1;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU, col = lc < NU;
        const int sx = (N + 1) * NX, su = N * NU, cj = col ? lc : 0;
        unsigned tfull[NTR], tcomp[NTR], poffs[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int a_ = 16 * ti + lc, a = a_ < 4 * RG ? a_ : 4 * RG - 1;
            tfull[ti] = (unsigned)((a >> 2) * TV * 64 + (a & 3) * 16 + lr);
            tcomp[ti] = (unsigned)(RG * TV * 64 + (a >> 2) * O::CT + (a & 3) * LV + lr);
            poffs[ti] = O::goff(0, ti, lr, lc);
        }
        for (int e = lane; e < NU * NX; e += NT) {
            const int j = e / NX;
            Ydx[j * sx + (e - j * NX)] = 0.0, Ydnu[j * sx + (e - j * NX)] = 0.0;
        }
        // -kff_0 of solve lc: column lc of R_0^-1, in the control slots
        const double k0 = (padl && col) ? minv2[4 * lr + cj] : 0.0;
        double w[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) w[rg] = 0.0;
        w[GQ] = k0;
        double nGt[D][NTR][RG], nP[D][RG][NTR];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks)
                        nGt[d][ti][ks] = G2[k * O::GSZ + (ks / 4 < TV ? tfull[ti] + (ks / 4) * 64 : tcomp[ti]) + 4 * (ks % 4)];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) nP[d][rg][ti] = P2[k * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
                });
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                d4_t acc[NTR], acc2[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = d4_t{0.0, 0.0, 0.0, 0.0}, acc2[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
                acc[GQ / 4][GQ % 4] = k == 0 ? k0 : 0.0;          // [b; -kff]: only -kff_0
                w[GQ] = padl ? acc[GQ / 4][GQ % 4] : w[GQ];
                double Gk[NTR][RG], Pk[RG][NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks) Gk[ti][ks] = nGt[d][ti][ks];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = nP[d][rg][ti];
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) {
                        acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ti][ks], w[ks], acc[ti], 0, 0, 0);
                        acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
                    }
                if (col)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        if constexpr (rg == GQ)
                            Ydx[padl ? (int)(Ydu.off - Ydx.off) + cj * su + k * NU + lr : cj * sx + (k + 1) * NX + om_xr<rg>(lr)] = v;
                        else
                            Ydx[rok ? cj * sx + (k + 1) * NX + om_xr<rg>(lr) : NU * sx] = v;
                        Ydnu[(rok && k > 0) ? cj * sx + k * NX + om_xr<rg>(lr) : NU * sx] = acc2[rg / 4][rg % 4];
                    });
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) w[rg] = acc[rg / 4][rg % 4];
            });
        {   // terminal multiplier step: Dnu_N = P_N dx_N
            d4_t acc2[NTR];
            double Pk[RG][NTR];
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = P2[N * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
            });
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti) acc2[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
            w[GQ] = padl ? 0.0 : w[GQ];
#pragma unroll
            for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
            if (col)
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                    Ydnu[rok ? cj * sx + N * NX + om_xr<rg>(lr) : NU * sx] = acc2[rg / 4][rg % 4];
                });
        }
        wave_sync();
    }

    // ---- multipliers of the dynamics at the end of a QP, by ONE costate sweep instead of Dnu_k = p_k + P_k Dx_k in every
    // interior-point iteration (which made the factor sweep stream P_k out and the corrector's forward sweep stream it back in: 9 KB
    // of the 32 KB a stage moved per iteration at n_mass 5).  No step of the iteration uses nuq — the Riccati direction gives Dx, Du,
    // and the bound rows give the step length — it is only an output, and the x rows of the QP's stationarity residual tie it to
    // what the iteration does carry:
    //     rg_x,k = q_x,k + (H dv_k)_x + A_k' nuq_{k+1} - nuq_k -+ lam_x,k        (rg: kept current by the (1 - alpha) scaling)
    // so  nuq_k = [q + H dv -+ lam - rg]_x,k + A_k' nuq_{k+1},  nuq_N = [..]_x,N : the same numbers as the accumulated steps, to
    // rounding.  H dv for all stages is three batches of MFMAs (16 stages per batch as the 16 columns of the B operand), the sweep a
    // chain of MFMAs on [B A]_k as it lies in the workspace.
    MPCRL_DI void costate_nu() {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NT_ = O::NT, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        const int ne = (N + 1) * NW;
        double *const lc_ = lds + Cfg::oBig, *const ltab = lc_ + (N + 1) * O::HBS;      // the stage vectors c_k (Omega order), the Hessian table
        // Hessian table in the register layout (= its A-operand layout: H is symmetric)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < NTR; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < NW && c < NW;
                ltab[(rg * NTR + tj) * 64 + lane] = in ? M::hess(false, O::nat(in ? e : 0), O::nat(in ? c : 0), th) : 0.0;
            }
        wave_sync();
        for (int b0 = 0; b0 <= N; b0 += 16) {
            const int st = b0 + lc, stc = st <= N ? st : N;      // this lane's stage (column lc of the batch)
            double op[RG];
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                double v;
                if constexpr (rg == GQ) {      // the control slots of the group take du (none at the terminal stage)
                    const double t_ = WsArr{dx.base, padl ? du.off + (unsigned)((stc < N ? stc : N - 1) * NU + lr) : dx.off + (unsigned)(stc * NX + om_xr<rg>(lr))}[0];
                    v = (padl && stc >= N) ? 0.0 : t_;
                } else
                    v = dx[stc * NX + (rok ? om_xr<rg>(lr) : 0)];
                op[rg] = rok ? v : 0.0;
            });
            d4_t y[NTR];
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti) {
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ltab[(ks * NTR + ti) * 64 + lane], op[ks], acc, 0, 0, 0);
                y[ti] = acc;
            }
            const double cks = ck(stc);
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                const int e = stc * NW + (rok ? om_nat<rg>(lr) : 0), i = rok ? om_nat<rg>(lr) : NU;
                double v = fma(cks, y[rg / 4][rg % 4], q[e]) - this->rg[e];
                if (has(0, stc, i)) v -= lam[e];
                if (has(1, stc, i)) v += lam[ne + e];
                if (st <= N) lc_[st * O::HBS + 4 * rg + lr] = rok ? v : 0.0;
            });
        }
        wave_sync();
        // the chain: nuq_k = c_k + A_k' nuq_{k+1}  (operand rows: next state, pad rows 0; columns: the state slots of stage k)
        int colnat[NTR];
#pragma unroll
        for (int tj = 0; tj < NTR; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const int rbase = lr * NW;
        d4_t R[NTR];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                double v = 0.0;
                if constexpr (rg < RG) v = lc_[N * O::HBS + 4 * rg + lr];
                R[ti][r] = v;
            });
        });
        auto store_nu = [&](int k) {
            if (lc == 0)
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                    nuq[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = R[rg / 4][rg % 4];
                });
        };
        store_nu(N);
        double nA[D][RG][NTR];
        staged_loop<D>(
            N - 1,     // stages N - 1 .. 1 (the multiplier of the initial condition is not an iterate)
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) nA[d][rg][tj] = BA[k * BAS + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW];
                });
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                double Ak[RG][NTR], vop[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) {
                        double v = nA[d][rg][tj];
                        if constexpr (rg == GQ) v = padl ? 0.0 : v;
                        if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                        Ak[rg][tj] = v;
                    }
                    vop[rg] = R[rg / 4][rg % 4];
                });
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ti][r] = 4 * ti + r < RG ? lc_[k * O::HBS + 4 * (4 * ti + r < RG ? 4 * ti + r : 0) + lr] : 0.0;
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[ks][ti], vop[ks], acc[ti], 0, 0, 0);
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) R[ti] = acc[ti];
                // the control slots of the result (B' nu) are not part of the vector: zero, so that the next operand's pad entries are
                R[GQ / 4][GQ % 4] = padl ? 0.0 : R[GQ / 4][GQ % 4];
                store_nu(k);
            });
        (void)NT_;
        wave_sync();
    }

    // ---- the interior-point iteration of qp_solve (below) with the BOUND ROWS IN REGISTERS (round 4; at most 128 rows: two per lane — the chain
    // problems bound the controls only, 3 x 40 rows).  Every row phase of qp_solve below is a pass over the rows through the
    // workspace: multipliers, slacks, the row's entry of the iterate and of the direction — a global-memory round trip (~2 us with
    // the chip streaming) per phase, ~27 of them per iteration, one lane-pass each.  Here a lane keeps its rows' (lam, t, aff, value,
    // residual entry) for the whole QP; what the sweeps need (barrier diagonal, modified gradient at the rows) is stored, the
    // direction at the rows is the one load per sweep, and the dense vector updates of an iteration are one fused pass.
    template <class HS>
    MPCRL_DI bool qp_solve_rows(HS &hs, const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu, bool rg_ready) {
        const bool warm = warm_mu > 0.0;
        const int ne = (N + 1) * NW;
        constexpr bool MERGED = MPCRL_CHAIN_MERGE_CALLS != 0 && std::is_same<HS, HessConst<M>>::value;
        for (int e = lane; e < (N + 1) * NX; e += NT) dx[e] = e < NX ? x0[e] - X[e] : 0.0, nuq[e] = warm ? NUv[e] : 0.0;
        for (int e = lane; e < N * NU; e += NT) du[e] = (qmode && e < NU) ? u0f[e] - U[e] : 0.0;
        wave_sync();
        // ---- this lane's rows
        bool on[2], hs_[2][2];
        int re_[2];
        unsigned doff[2], Doff[2];      // where the row's entry of (dx | du) and (Dx | Du) sits, relative to dx / Dx
        double lb_[2], ub_[2], v0[2], dvq[2], rgr[2], lm[2][2], tt_[2][2], af[2][2];
        double cnt = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r_ = lane + 64 * j;
            on[j] = r_ < nrows;
            int k = 0, i = 0;
            if (on[j]) row_of(r_, k, i);
            re_[j] = k * NW + i;
            lb_[j] = lbv(k, i), ub_[j] = ubv(k, i);
            hs_[j][0] = on[j] && has(0, k, i), hs_[j][1] = on[j] && has(1, k, i);
            const bool isu = i < NU;
            doff[j] = isu ? (du.off - dx.off) + (unsigned)((k < N ? k : 0) * NU + i) : (unsigned)(k * NX + i - NU);
            Doff[j] = isu ? (Du.off - Dx.off) + (unsigned)((k < N ? k : 0) * NU + i) : (unsigned)(k * NX + i - NU);
            v0[j] = on[j] ? vc(k, i) : 0.0;
            dvq[j] = on[j] ? dx[(int)doff[j]] : 0.0;
            rgr[j] = on[j] ? rg[re_[j]] : 0.0;
            const double v = v0[j] + dvq[j];
            double drg = 0.0;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                lm[j][sd] = 0.0, tt_[j][sd] = 1.0, af[j][sd] = 0.0;
                if (hs_[j][sd]) {
                    cnt += 1.0;
                    const double l_old = LAM(sd, re_[j]), sl = sd ? ub_[j] - v : v - lb_[j];
                    double l, t1;
                    if (warm) {
                        l = l_old, t1 = fmax(sl, TT(sd, re_[j]));
                        if (l * t1 < warm_mu) {
                            if (l >= t1)
                                t1 = warm_mu / l;
                            else
                                l = warm_mu / t1;
                        }
                    } else {
                        t1 = fmax(sl, IPM_T_MIN);
                        l = IPM_MU0 / t1;
                    }
                    lm[j][sd] = l, tt_[j][sd] = t1;
                    drg += sd ? l - l_old : l_old - l;
                }
            }
            if (rg_ready && on[j] && !skipc(k, i) && !fixedc(k, i)) {
                rgr[j] += drg;
                rg[re_[j]] = rgr[j];
            }
        }
        auto store_rows = [&]() {      // multipliers and slacks back to the workspace (next QP's warm start, the kernel's write-out)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) LAM(sd, re_[j]) = lm[j][sd], TT(sd, re_[j]) = tt_[j][sd];
        };
        const double n_rows = wave_sum(cnt);
        if (!rg_ready) store_rows();      // (qp_residuals reads lam from the workspace)
        wave_sync();
        bool ok = false, stepped = false;
        double rlin = wave_max(rg_ready ? qp_start_residuals() : qp_residuals_call(ctx()));
        if (!rg_ready) {
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] = on[j] ? rg[re_[j]] : 0.0;
        } else {      // the stage-0 terms of a warm solve from a new state may have touched this lane's entries
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] = (on[j] && re_[j] < NW) ? rg[re_[j]] : rgr[j];
        }
        // rt = rg, Dg = 0 once: the rows rewrite their own entries in every pass, the other entries of rt follow rg in the fused update
        batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int e, double v) { rt[e] = v, Dg[e] = 0.0; });
        wave_sync();
        for (int it = 0;; ++it) {
            ph(7);
            double rloc = rlin, muloc = 0.0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double v = v0[j] + dvq[j];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) {
                        rloc = fmax(rloc, fabs(tt_[j][sd] - (sd ? ub_[j] - v : v - lb_[j])));
                        muloc = fma(lm[j][sd], tt_[j][sd], muloc);
                    }
            }
            const double rinf = wave_max(rloc);
            const double mu = n_rows > 0.0 ? wave_sum(muloc) / n_rows : 0.0;
            if (rinf <= tol_res && mu <= tol_mu) {
                ok = true;
                break;
            }
            if (it >= IPM_MAX_ITER || !(rinf < 1e300)) break;
            ++n_it;
            ph(0);
            double sigma_mu = 0.0, alpha = 1.0, dvr[2] = {0.0, 0.0};
            bool fail = false;
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double dg = 0.0, er = 0.0;
                    const double v = v0[j] + dvq[j];
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (hs_[j][sd]) {
                            const double l1 = lm[j][sd], t1 = tt_[j][sd];
                            const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                            const double rm = fma(l1, t1, pass ? af[j][sd] - sigma_mu : 0.0);
                            dg += l1 / t1;
                            er += (sd ? -1.0 : 1.0) * (rm - l1 * rd1) / t1;
                        }
                    if (on[j]) {
                        if (pass == 0) Dg[re_[j]] = dg;
                        rt[re_[j]] = rgr[j] + er;
                    }
                }
                wave_sync();
                ph(1);
                if constexpr (MERGED) {
                    if (pass == 0) {
                        if (!pred_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                    } else
                        corr_call(ctx(), rt.off, rb.off);
                } else {
                    if (pass == 0) {
                        if (!factor_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                        ph(2);
                    } else {
                        backward_vec_call(ctx(), rt.off);
                        ph(3);
                    }
                    forward_call(ctx(), rb.off);
                }
                ph(4);
#pragma unroll
                for (int j = 0; j < 2; ++j) dvr[j] = on[j] ? Dx[(int)Doff[j]] : 0.0;      // the direction at the rows: the one load of the pass
                double amax = 1.0, muaff = 0.0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (hs_[j][sd]) {
                            const double l1 = lm[j][sd], t1 = tt_[j][sd];
                            const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                            const double rm = fma(l1, t1, pass ? af[j][sd] - sigma_mu : 0.0);
                            const double dt1 = -rd1 + (sd ? -dv : dv);
                            const double dl1 = (-rm - l1 * dt1) / t1;
                            if (dl1 < 0.0) amax = fmin(amax, -l1 / dl1);
                            if (dt1 < 0.0) amax = fmin(amax, -t1 / dt1);
                        }
                }
                amax = -wave_max(-amax);
                if (pass == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd)
                            if (hs_[j][sd]) {
                                const double l1 = lm[j][sd], t1 = tt_[j][sd];
                                const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                                const double dt1 = -rd1 + (sd ? -dv : dv);
                                const double dl1 = (-l1 * t1 - l1 * dt1) / t1;
                                muaff = fma(fma(amax, dl1, l1), fma(amax, dt1, t1), muaff);
                                af[j][sd] = dl1 * dt1;
                            }
                    }
                    const double mu_aff = n_rows > 0.0 ? wave_sum(muaff) / n_rows : 0.0;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                } else
                    alpha = fmin(1.0, fmax(IPM_FRAC, frac_fixed ? 0.0 : 1.0 - mu) * amax);   // fraction to the boundary -> 1 as mu -> 0
            }
            ph(5);
            if (fail) break;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) {
                        const double l1 = lm[j][sd], t1 = tt_[j][sd];
                        const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                        const double rm = fma(l1, t1, af[j][sd] - sigma_mu);
                        const double dt1 = -rd1 + (sd ? -dv : dv);
                        const double dl1 = (-rm - l1 * dt1) / t1;
                        lm[j][sd] = fma(alpha, dl1, l1);
                        tt_[j][sd] = fma(alpha, dt1, t1);
                    }
                dvq[j] = fma(alpha, dvr[j], dvq[j]);      // (the same fma the dense update below applies to the entry)
            }
            // one fused pass: dx += alpha Dx, du += alpha Du, rg *= (1 - alpha) (and rt = rg), rb *= (1 - alpha)
            const double om = 1.0 - alpha;
            struct Upd {
                double a, b, c, d, e, f;
            };
            batched_pass<4>(ne, lane,
                            [&](int e) {
                                const int ex = e < (N + 1) * NX ? e : 0, eu = e < N * NU ? e : 0, eb = e < N * NX ? e : 0;
                                return Upd{rg[e], Dx[ex], dx[ex], Du[eu], du[eu], rb[eb]};
                            },
                            [&](int e, const Upd &v) {
                                const double g = om * v.a;
                                rg[e] = g, rt[e] = g;
                                if (e < (N + 1) * NX) dx[e] = fma(alpha, v.b, v.c);
                                if (e < N * NU) du[e] = fma(alpha, v.d, v.e);
                                if (e < N * NX) rb[e] = om * v.f;
                            });
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] *= om;
            rlin *= om;
            stepped = true;
            wave_sync();
        }
        store_rows();
        wave_sync();
        if (stepped) costate_call(ctx());      // nuq of the point the iteration ended at (no iteration: the warm multipliers stand)
        return ok;
    }

    // ---- Mehrotra predictor-corrector on the QP of the current linearisation (hard bounds) --------------------
    // rg_ready: rg holds q -+ lam + [B A]' NUv - [0; NUv] of this linearisation (round_start left it there) and the QP starts from
    // nuq = NUv (warm) or from NUv = 0 (cold): its starting residual needs no pass over the [B A]_k
    template <class HS>
    MPCRL_DI bool qp_solve(HS &hs, const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu, bool rg_ready = false) {
        if (nrows <= 128) return qp_solve_rows(hs, x0, u0f, n_it, warm_mu, tol_res, tol_mu, rg_ready);
        // more rows than two per lane (state bounds set through mpcrl_set_bounds): the row phases as passes over the workspace
        const bool warm = warm_mu > 0.0;
        const int ne = (N + 1) * NW;
        for (int e = lane; e < (N + 1) * NX; e += NT) dx[e] = e < NX ? x0[e] - X[e] : 0.0, nuq[e] = warm ? NUv[e] : 0.0;
        for (int e = lane; e < N * NU; e += NT) du[e] = (qmode && e < NU) ? u0f[e] - U[e] : 0.0;
        wave_sync();
        double cnt = 0.0;
        for (int r_ = lane; r_ < nrows; r_ += NT) {
            int k, i;
            row_of(r_, k, i);
            const int e = k * NW + i;
            const double v = vc(k, i) + dvc(dx, du, k, i);
            double drg = 0.0;      // the multipliers of this row change: so does its entry of the stationarity residual
            for (int sd = 0; sd < 2; ++sd)
                if (has(sd, k, i)) {
                    cnt += 1.0;
                    const double l_old = LAM(sd, e);
                    double l_new;
                    if (warm) {
                        double l = l_old, tt = fmax(bslack(sd, k, i, v), TT(sd, e));
                        if (l * tt < warm_mu) {
                            if (l >= tt)
                                tt = warm_mu / l;
                            else
                                l = warm_mu / tt;
                        }
                        LAM(sd, e) = l, TT(sd, e) = tt;
                        l_new = l;
                    } else {
                        const double tt = fmax(bslack(sd, k, i, v), IPM_T_MIN);
                        TT(sd, e) = tt;
                        l_new = IPM_MU0 / tt;
                        LAM(sd, e) = l_new;
                    }
                    drg += sd ? l_new - l_old : l_old - l_new;
                }
            if (rg_ready && !skipc(k, i) && !fixedc(k, i)) rg[e] += drg;
        }
        const double n_rows = wave_sum(cnt);
        wave_sync();
        bool ok = false, stepped = false;
        double rlin = 0.0;
        for (int it = 0;; ++it) {
            ph(7);
            // The residuals of the LINEAR equations (dynamics rb, stationarity rg) are evaluated once per QP: a step of length alpha
            // along a direction that solves the Newton system takes them to (1 - alpha) times their value, exactly — they are
            // scaled at the end of the iteration instead of being re-evaluated (a sweep over all [B A]_k: 161 KB per instance at
            // n_mass 5, 10 % of the kernel).  Only the bound rows below depend on the step nonlinearly (complementarity).
            if (it == 0) rlin = wave_max(rg_ready ? qp_start_residuals() : qp_residuals_call(ctx()));
            double rloc = rlin, muloc = 0.0;
            for (int r_ = lane; r_ < nrows; r_ += NT) {
                int k, i;
                row_of(r_, k, i);
                const int e = k * NW + i;
                const double v = vc(k, i) + dvc(dx, du, k, i);
                for (int sd = 0; sd < 2; ++sd)
                    if (has(sd, k, i)) {
                        rloc = fmax(rloc, fabs(TT(sd, e) - bslack(sd, k, i, v)));
                        muloc = fma(LAM(sd, e), TT(sd, e), muloc);
                    }
            }
            const double rinf = wave_max(rloc);
            const double mu = n_rows > 0.0 ? wave_sum(muloc) / n_rows : 0.0;
            if (rinf <= tol_res && mu <= tol_mu) {
                ok = true;
                break;
            }
            if (it >= IPM_MAX_ITER || !(rinf < 1e300)) break;
            ++n_it;
            ph(0);
            double sigma_mu = 0.0, alpha = 1.0;
            bool fail = false;
            for (int pass = 0; pass < 2; ++pass) {
                // barrier diagonal + modified gradient: rt = rg everywhere, corrected on the bounded rows
                batched_pass<8>(ne, lane, [&](int e) { return rg[e]; },
                                [&](int e, double v) {
                                    rt[e] = v;
                                    if (pass == 0) Dg[e] = 0.0;
                                });
                wave_sync();
                for (int r_ = lane; r_ < nrows; r_ += NT) {
                    int k, i;
                    row_of(r_, k, i);
                    const int e = k * NW + i;
                    double dg = 0.0, er = 0.0;
                    const double v = vc(k, i) + dvc(dx, du, k, i);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has(sd, k, i)) {
                            const double l1 = LAM(sd, e), t1 = TT(sd, e);
                            const double rd1 = t1 - bslack(sd, k, i, v);
                            const double rm = fma(l1, t1, pass ? AFF(sd, e) - sigma_mu : 0.0);
                            dg += l1 / t1;
                            er += (sd ? -1.0 : 1.0) * (rm - l1 * rd1) / t1;
                        }
                    if (pass == 0) Dg[e] = dg;
                    rt[e] = rg[e] + er;
                }
                wave_sync();
                ph(1);
                if (pass == 0) {
                    if (!factor_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                    ph(2);
                } else {
                    backward_vec_call(ctx(), rt.off);
                    ph(3);
                }
                forward_call(ctx(), rb.off);
                ph(4);
                double amax = 1.0, muaff = 0.0;
                for (int r_ = lane; r_ < nrows; r_ += NT) {
                    int k, i;
                    row_of(r_, k, i);
                    const int e = k * NW + i;
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
                amax = -wave_max(-amax);
                if (pass == 0) {
                    for (int r_ = lane; r_ < nrows; r_ += NT) {
                        int k, i;
                        row_of(r_, k, i);
                        const int e = k * NW + i;
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
                    const double mu_aff = n_rows > 0.0 ? wave_sum(muaff) / n_rows : 0.0;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                    wave_sync();
                } else
                    alpha = fmin(1.0, fmax(IPM_FRAC, frac_fixed ? 0.0 : 1.0 - mu) * amax);   // fraction to the boundary -> 1 as mu -> 0
            }
            ph(5);
            if (fail) break;
            for (int r_ = lane; r_ < nrows; r_ += NT) {
                int k, i;
                row_of(r_, k, i);
                const int e = k * NW + i;
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
            wave_sync();
            batched_pass<8>((N + 1) * NX, lane, [&](int e) { return Pair2{Dx[e], dx[e]}; },
                            [&](int e, const Pair2 &v) { dx[e] = fma(alpha, v.a, v.b); });
            stepped = true;
            batched_pass<2>(N * NU, lane, [&](int e) { return Pair2{Du[e], du[e]}; },
                            [&](int e, const Pair2 &v) { du[e] = fma(alpha, v.a, v.b); });
            {
                const double om = 1.0 - alpha;
                batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int e, double v) { rg[e] = om * v; });
                batched_pass<8>(N * NX, lane, [&](int e) { return rb[e]; }, [&](int e, double v) { rb[e] = om * v; });
                rlin *= om;
            }
            wave_sync();
        }
        if (stepped) costate_call(ctx());      // nuq of the point the iteration ended at (no iteration: the warm multipliers stand)
        return ok;
    }

    // ---- phase calls.  The big phases are real (non-inlined) functions: each gets a register allocation of its own, so the
    // operands of one phase are never spilled on behalf of another (inlined into one body, the interior-point loop carried
    // hundreds of hoisted loop invariants through every phase, and each reload from scratch is an s_waitcnt vmcnt(0) that also
    // drains the streaming stores).  What travels is a CONTEXT of 16 dwords (argument registers): workspace, parameters, iterate,
    // LDS addresses, horizon.  Until round 3 the solver itself travelled by value — ~140 dwords per lane, i.e. 35 KB per wavefront
    // written to and read back from scratch memory at every call (4.7 KB per lane of frame, a large part of the kernel's HBM
    // traffic beyond its streamed factors); every field of it is a function of the context: the workspace arrays are offsets of one
    // base (LargeLayout), the tile indices functions of the lane, the row counts three words the set-up left in LDS.
    MPCRL_DI static unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
    template <class T>
    MPCRL_DI static T *uni_global(T *ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = rfl((unsigned)v), hi = rfl((unsigned)(v >> 32));
        typedef __attribute__((address_space(1))) T GT;
        return (T *)(GT *)(((unsigned long long)hi << 32) | lo);
    }
    template <class T>
    MPCRL_DI static T *uni_lds(T *ptr) {
        typedef __attribute__((address_space(3))) T LT;
        return (T *)(LT *)(unsigned long)rfl((unsigned)(unsigned long long)ptr);
    }
    MPCRL_DI void uni(WsArr &a_) const { a_.base = a_.base ? BA.base : nullptr, a_.off = rfl(a_.off); }
    struct Ctx {
        double *w, *X, *U;
        const double *th, *xs;
        double *lds;
        int *sidx;
        int N, qmode;
#ifdef MPCRL_PROFILE_PHASES
        unsigned long long *ph_lds;
#endif
    };
    MPCRL_DI Ctx ctx() const {
        Ctx c;
        c.w = (double *)BA.base, c.X = X, c.U = U, c.th = th, c.xs = xs, c.lds = lds, c.sidx = sidx, c.N = N, c.qmode = qmode ? 1 : 0;
#ifdef MPCRL_PROFILE_PHASES
        c.ph_lds = ph_lds;
#endif
        return c;
    }
    // the solver of a phase, rebuilt from the context; wave-uniform values end up in scalar registers (readfirstlane on entry,
    // everything derived from them is scalar arithmetic), pointers get their address spaces back
    MPCRL_DI static ChainSolver from_ctx(const Ctx &c) {
        ChainSolver S(uni_global(c.xs), (int)rfl((unsigned)c.N), (int)threadIdx.x);
        S.qmode = rfl((unsigned)c.qmode) != 0;
        S.th = uni_global(c.th), S.X = uni_global(c.X), S.U = uni_global(c.U);
        S.bind_workspace(uni_global(c.w), LargeLayout<M>(S.N));
        S.setup_lane(uni_lds(c.lds), uni_lds(c.sidx), true);
#ifdef MPCRL_PROFILE_PHASES
        S.ph_lds = uni_lds(c.ph_lds);
        S.ph_t = clock64();
#endif
        return S;
    }
    MPCRL_DI WsArr arr(unsigned off) const { return WsArr{BA.base, rfl(off)}; }
    struct RoundStart {
        double cost, res[4];
    };
    __device__ MPCRL_PHASE_FN static RoundStart round_start_call(Ctx c, const double *x0, const double *u0f) {
        ChainSolver S = from_ctx(c);
        x0 = uni_global(x0), u0f = u0f ? uni_global(u0f) : nullptr;
        RoundStart o;
        o.cost = S.round_start(x0, u0f, o.res);
        return o;
    }
    // the whole start of an SQP round as ONE call: parameter / multiplier staging, point pass, direction pass, round_start.  As
    // three calls their prologues and epilogues moved ~70 KB per wavefront and round through scratch (MPCRL_CHAIN_MERGE_CALLS).
    __device__ MPCRL_PHASE_FN static RoundStart round_call(Ctx c, const double *x0, const double *u0f, double h, int steps) {
        ChainSolver S = from_ctx(c);
        using DC_ = DirCfg<M>;
        double *const big = S.lds + Cfg::oBig;
        const int N_ = S.N, lane_ = S.lane;
        for (int e = lane_; e < M::NTD; e += NT) big[DC_::CO + e] = S.th[M::td_index(e)];
        if constexpr (Cfg::FUSE_GT)
            batched_pass<8>((N_ + 1) * NX, lane_, [&](int e) { return S.NUv[e]; }, [&](int e, double v) { big[DC_::CO + M::NTD + e] = v; });
        wave_sync();
        if (lane_ < N_)
            chain_point_body<M, false, true>(S.X, S.U, big + DC_::CO, (double *)S.BA.base, N_, lane_, h, steps,
                                             big + DC_::CO + M::NTD + (Cfg::FUSE_GT ? (N_ + 1) * NX : 0));
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        chain_dir_body<M>(S.th, (double *)S.BA.base, big, N_, lane_, h, steps);
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        RoundStart o;
        o.cost = S.round_start(x0, u0f, o.res);
        return o;
    }
    __device__ MPCRL_PHASE_FN static double qp_residuals_call(Ctx c) {
        ChainSolver S = from_ctx(c);
        return S.qp_residuals2();
    }
    // hex_off: workspace offset of the exact Hessian blocks (HessGlobal); the constant Hessian (HessConst) is rebuilt from theta
    template <class HS>
    __device__ MPCRL_PHASE_FN static bool factor_call(Ctx c, unsigned hex_off, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        HS hs;
        hs.init(S, hex_off);
        return S.template factor2<HS, !std::is_same<HS, HessConst<M>>::value>(hs, S.arr(g_off), S.arr(bb_off));
    }
    __device__ MPCRL_PHASE_FN static void backward_vec_call(Ctx c, unsigned g_off) {
        ChainSolver S = from_ctx(c);
        S.backward_vec2(S.arr(g_off));
    }
    __device__ MPCRL_PHASE_FN static void forward_sens_call(Ctx c, unsigned ydx, unsigned ydu, unsigned ydnu) {
        ChainSolver S = from_ctx(c);
        S.forward2_sens(S.arr(ydx), S.arr(ydu), S.arr(ydnu));
    }
    __device__ MPCRL_PHASE_FN static void costate_call(Ctx c) {
        ChainSolver S = from_ctx(c);
        S.costate_nu();
    }
    // predictor and corrector as one call each (round-4 sweeps of the SQP only).  Every phase call saves and restores the callee-saved
    // half of the registers its body uses — 29 KB per wavefront for the factor sweep, ~20 KB for a vector sweep, through scratch, i.e.
    // HBM traffic at 1024 resident wavefronts: ~1.9 GB written and read back per step of 1024 solves at n_mass 5 with four calls per
    // interior-point iteration.  The time is the same either way (measured: 8.68 vs 8.69 ms), the bytes are not.
    template <class HS>
    __device__ MPCRL_PHASE_FN static bool pred_call(Ctx c, unsigned hex_off, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        HS hs;
        hs.init(S, hex_off);
        const bool ok = S.template factor2<HS, false>(hs, S.arr(g_off), S.arr(bb_off));
        S.forward2(S.arr(bb_off));
        return ok;
    }
    __device__ MPCRL_PHASE_FN static void corr_call(Ctx c, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        S.backward_vec2(S.arr(g_off));
        S.forward2(S.arr(bb_off));
    }
    __device__ MPCRL_PHASE_FN static void forward_call(Ctx c, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        S.forward2(S.arr(bb_off));
    }

    MPCRL_DI void bind_workspace(double *w, const LargeLayout<M> &lay) {
        auto at = [&](size_t o) { return WsArr{(char *)w, (unsigned)o}; };
        BA = at(lay.BA), r = at(lay.r), q = at(lay.q), dx = at(lay.dx), du = at(lay.du), nuq = at(lay.nuq);
        Dx = at(lay.Dx), Du = at(lay.Du), rg = at(lay.rg), rb = at(lay.rb), rt = at(lay.rt), Dg = at(lay.Dg);
        lam = at(lay.lamw), t = at(lay.tw), aff = at(lay.aff), p = at(lay.p), kff = at(lay.kff), NUv = at(lay.ynu), state = at(lay.state);
        G2 = at(lay.G2), P2 = at(lay.P2), hb2 = at(lay.hb2), minv2 = at(lay.minv2), mvu2 = at(lay.mvu2);
    }
};

}  // namespace mpcrl
