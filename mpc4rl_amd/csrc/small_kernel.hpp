// small_kernel.hpp — the register-resident SQP / Riccati-IPM / adjoint-sensitivity kernel for small OCPs
// (cartpole nx=4 nu=1, linear system nx=2 nu=1) on gfx950.
//
// Replaces, for a whole batch at once, what the reference does per instance through
//   ocp_solver.solve()            rlmpc/mpc/common/mpc.py:42,79,195     (acados SQP + HPIPM, not vendored)
//   update_nlp(): dL_dp, dpi_dp   rlmpc/mpc/nlp.py:1399-1424            (dense 600x600 Jacobian + SuperLU)
//
// Mapping (MI355X-first, not a translation of anything):
//   * one wavefront (64 lanes) per workgroup; ONE LANE PER SHOOTING STAGE: lane k of an instance owns
//     x_k, u_k, the multiplier of the dynamics arriving at stage k, A_k, B_k, the bound multipliers of stage k
//     and the Riccati factors of stage k, all in VGPRs.  floor(64/(N+1)) instances share a wave
//     (cartpole N=20: 3 instances = 63 lanes).
//   * everything that is independent per stage (RK4 + forward-mode Jacobians, residuals, barrier terms,
//     step recovery, fraction-to-boundary, exact-Hessian second-order sweeps) runs stage-parallel with no
//     memory traffic at all;
//   * the Riccati recursion is the only serial part: (P, p) travel lane k+1 -> lane k with cross-lane
//     moves, the forward sweep sends the state step lane k -> lane k+1;
//   * per-instance reductions (residual norms, mu, step length) are segmented wave reductions.
//   * HBM traffic is the algorithmic minimum: x0 in, iterate in/out, results out.
//
// The interior-point iteration (initial point, Mehrotra predictor-corrector, single step length,
// fraction to the boundary max(0.995, 1 - mu), stopping rule) is the one documented in DESIGN.md; its constants are below.
#pragma once
#include "models_dev.hpp"

namespace mpcrl {

#ifdef MPCRL_PROFILE_PHASES
static __device__ unsigned long long g_phase_ticks[16];   // (one copy per translation unit: mpcrl_debug_phases sums them)
#define PHW(i) do { unsigned long long n_ = clock64(); phw[i] += n_ - pht; pht = n_; } while (0)
#define PHW_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_ticks[i_], phw[i_]); } while (0)
#else
#define PHW(i)
#define PHW_FLUSH()
#endif


constexpr int IPM_MAX_ITER = 60;
constexpr double IPM_TOL_RES = 1e-9, IPM_TOL_MU = 1e-11, IPM_T_MIN = 1e-1, IPM_MU0 = 1.0, IPM_FRAC = 0.995;
constexpr double NO_BOUND = 1e29;
// warm start of the interior point method from the previous QP: mu_w = clamp(C * step^2, MIN, MAX)   
constexpr double IPM_WARM_C = 1e-4, IPM_WARM_MIN = 1e-10, IPM_WARM_MAX = 3e-2;
// inexact SQP: QP tolerances follow the NLP residual r (tol_res = clamp(C r^2, TOL_RES, CAP), tol_mu = clamp(C r^2 / 100, TOL_MU, CAP / 10));
// convergence is only declared after a QP solved to the tight tolerances   
constexpr double IPM_ADAPT_C = 1e1, IPM_ADAPT_CAP = 3e-2;
// Where the predictor step alone cuts the complementarity by more than two orders of magnitude (sigma = (mu_aff / mu)^3 below this)
// it is the Newton step of a QP that is all but solved: it is taken and the corrector's two vector sweeps are not run.  Models opt in
// (M::SKIP_CORRECTOR: the cartpole — a third of its interior-point iterations, none more needed).
constexpr double IPM_SKIP_SIGMA = 3e-6;
#ifndef MPCRL_MX_SLOTS
#define MPCRL_MX_SLOTS 64        // stage slots of the matrix-layout sweeps in LDS (one per lane)
#endif
// Adjoint (sensitivity) solve: the stiffness lam / t of an active bound row is capped.  A row whose slack the interior point took to
// 1e-18 pins its coordinate either way (the answer moves by O(1 / stiffness)), but 1e19 on the diagonal of a STATE block costs the
// Riccati recursion all sixteen digits of the entries next to it (against a dense pivoted solve on the hardest test instances:
// cap 1e8 -> 2e-7, 1e9 -> 1e-7 (bias 1e-8), 1e10 -> 1e-6, 1e12 -> 9e-5, 1e14 -> 3e-2, none -> 5e-1).  Round 6: with an active STATE
// bound the bias is larger than on those instances (c / W with c ~ 3.6e3: 3.6e-6 at 1e9) — removed by Richardson extrapolation in
// the cap where it occurs (SmallSolver::sensitivities, matrix-layout models).
constexpr double SENS_W_MAX = 1e9;

struct SmallArgs {
    int B;                 // instances
    int flags;             // MPCRL_* solve flags
    int theta_stride;      // 0 = shared theta, np = per instance
    const int *perm;       // packing order: slot i of the launch works on instance perm[i] (null = identity)
    const int *cold;       // [B] or null: non-zero = this instance ignores its stored iterate (per-instance MPCRL_COLD)
    const double *x0;      // [B, nx]
    const double *u0fix;   // [B, nu] or null
    const double *theta;   // [np] or [B, np]
    double *X, *U, *PI, *BND, *RES;   // iterate workspace (in/out), layouts of mpcrl_get_iterate
    double *LAG;           // [B] Lagrangian of the mirror at the returned iterate (mpcrl_get_lagrangian)
    double *u0_out, *V, *dV, *dpi;
    int *status, *iters;
};

// ---- cross-lane helpers (wave64) -----------------------------------------------------------------
// One-lane shifts across the whole wave as DPP moves (gfx9 wave_shl:1 / wave_shr:1): a VALU op instead of the LDS-crossbar
// round trip of ds_bpermute that __shfl_down/__shfl_up compile to; same edge behaviour (the last / first lane keeps its own value).
MPCRL_DI double lane_dn(double v) {   // value of lane + 1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
MPCRL_DI double lane_up(double v) {   // value of lane - 1
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// 1/x from the hardware seed plus two Newton steps (~1 ulp); x is a strictly positive, normal number wherever this is used
// (slacks, multipliers, pivots), so the range / denormal handling of an IEEE division (~13 instructions) is not needed.
MPCRL_DI double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// lane select by a compile-time lane mask (bit l = lane l takes `yes`): two v_cndmask_b32 as inline assembly, so that it STAYS a
// select (a ternary on a lane-dependent condition with non-trivial arms may be turned into exec-mask branches)
MPCRL_DI double lane_select(unsigned long long mask, double yes, double no) {
    int yl = __double2loint(yes), yh = __double2hiint(yes), nl = __double2loint(no), nh = __double2hiint(no), rl, rh;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(rl) : "v"(nl), "v"(yl), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(rh) : "v"(nh), "v"(yh), "s"(mask));
    return __hiloint2double(rh, rl);
}
constexpr unsigned long long LANES_C0 = 0x1111111111111111ull, LANES_C1 = 0x2222222222222222ull, LANES_C01 = 0x3333333333333333ull;   // lane & 3 == 0 / == 1 / < 2
#ifndef MPCRL_V_ASMSEL
#define MPCRL_V_ASMSEL 1
#endif

#ifndef MPCRL_SMALL_SCAN
#define MPCRL_SMALL_SCAN 1      // vector sweeps of the stage-per-lane layout as parallel scans (SmallSolver::SCAN)
#endif
#ifndef MPCRL_IPM_SCALE_RES
#define MPCRL_IPM_SCALE_RES 1
#endif
// segmented reductions over the LPI lanes of one instance; result broadcast to all of its lanes
// SKIP (a model constant, M::SEG_SKIP): leave out the tree levels at which no lane has a partner (s >= lpi, wave-uniform).  One
// cross-lane round trip less per reduction on paper; measured with everything else equal it is 2.3 % SLOWER for the cartpole kernels
// (the branches change the schedule around the reductions) and 4 % faster for the linear system's, so each model states its own.
template <bool SKIP>
MPCRL_DI double seg_sum(double v, int k, int lpi, int base) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        if (SKIP && s >= lpi) continue;   // wave-uniform: no lane has a partner at this distance (a cross-lane round trip saved)
        const double o = __shfl_down(v, s);
        if (k + s < lpi) v += o;
    }
    return __shfl(v, base);
}
template <bool SKIP>
MPCRL_DI double seg_max(double v, int k, int lpi, int base) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        if (SKIP && s >= lpi) continue;
        const double o = __shfl_down(v, s);
        if (k + s < lpi) v = fmax(v, o);
    }
    return __shfl(v, base);
}
template <bool SKIP>
MPCRL_DI double seg_min(double v, int k, int lpi, int base) { return -seg_max<SKIP>(-v, k, lpi, base); }
// several reductions in one pass: the cross-lane moves of the different values overlap instead of queueing behind each other
template <int NMAX, int NSUM, bool SKIP>
MPCRL_DI void seg_reduce(double *mx, double *sm, int k, int lpi, int base) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        if (SKIP && s >= lpi) continue;
        double om[NMAX > 0 ? NMAX : 1], os[NSUM > 0 ? NSUM : 1];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) om[i] = __shfl_down(mx[i], s);
#pragma unroll
        for (int i = 0; i < NSUM; ++i) os[i] = __shfl_down(sm[i], s);
        if (k + s < lpi) {
#pragma unroll
            for (int i = 0; i < NMAX; ++i) mx[i] = fmax(mx[i], om[i]);
#pragma unroll
            for (int i = 0; i < NSUM; ++i) sm[i] += os[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) mx[i] = __shfl(mx[i], base);
#pragma unroll
    for (int i = 0; i < NSUM; ++i) sm[i] = __shfl(sm[i], base);
}

template <class M>
struct SmallSolver {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NTC = M::NTC, NP = M::NP;
    static constexpr int NPK = NX * (NX + 1) / 2, NLK = NU * (NU + 1) / 2;
    static constexpr bool SOFT = M::HAS_SOFT;
    MPCRL_DI static constexpr int sym(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

    const SmallSpec &sp;
    const int N, lpi, k, base, blkidx;   // blkidx: which instance slot of the wavefront (base / lpi)
    const bool term, first;
    bool qmode;
    double ck;                    // cost scaling c_k of this stage
    // the instance's differentiable / cost parameters: registers, or (M::DISCRETE, the linear-system model: 12 doubles that are read a
    // few times per SQP iteration and were simply parked in scratch) an LDS copy per instance slot of the wavefront
    template <int NN, bool LDS_>
    struct ThStore {
        double a[NN > 0 ? NN : 1];
        MPCRL_DI double operator[](int i) const { return a[i]; }
        MPCRL_DI void set(int i, double v) { a[i] = v; }
        MPCRL_DI const double *ptr() const { return a; }
        MPCRL_DI void bind(double *) {}
    };
    template <int NN>
    struct ThStore<NN, true> {
        double *p;
        MPCRL_DI double operator[](int i) const { return p[i]; }
        MPCRL_DI void set(int i, double v) { p[i] = v; }   // (every lane of the instance writes the same value)
        MPCRL_DI const double *ptr() const { return p; }
        MPCRL_DI void bind(double *q_) { p = q_; }
    };
    static constexpr bool TH_LDS = M::DISCRETE;
    static constexpr int TH_DOUBLES = NTD + (NTC > 0 ? NTC : 1);
    ThStore<NTD, TH_LDS> thd;
    ThStore<NTC, TH_LDS> thc;
    MPCRL_DI void bind_theta(double *slot_base) { thd.bind(slot_base), thc.bind(slot_base + NTD); }
    // NLP iterate of this stage
    double x[NX], u[NU], nu_[NX];   // nu_ = multiplier of x_k = F(x_{k-1},u_{k-1})   (k >= 1)
    // linearisation of the dynamics leaving this stage (k < N) and cost gradient
    double A[NX * NX], Bm[NX * NU], r[NX], q[NW];
    // inequality rows of this stage: [side 0 lower / 1 upper][coordinate of v = [u; x]]
    double lam[2][NW], t[2][NW], aff[2][NW];
    // slack state of the L1-soft bounds: one slot per coordinate the model's OCP can soften (M::NSOFT, M::soft_slot) — carried for all NW
    // coordinates it was 24 doubles of state per lane in the linear-system kernel for ONE soft coordinate, most of them spilled
    static constexpr int NS = M::NSOFT;
    MPCRL_DI static constexpr int ss(int i) { return M::soft_slot(i); }
    double s[2][NS], lams[2][NS], ts[2][NS], affs[2][NS];
    // QP iterate and Newton step
    double dx[NX], du[NU], nuq[NX], Dx[NX], Du[NU], Dnu[NX];
    // Riccati factors of this stage
    double K[NU * NX], Li[NLK], kff[NU], P[NPK], p[NX];
    double rg[NW], rb[NX], rt[NW], Dg[NW];
#ifdef MPCRL_PROFILE_PHASES
    unsigned long long phw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pht = 0;
#endif
    double hscale = 1.0;   // multiplies the Hs accessor of the Riccati stage (c_k for the SQP Hessian, 1 for the exact one)
    double n_rows_c = -1.0;   // bound rows per instance, counted by the first QP of the launch
    // the instance's x0 (and pinned u0 in Q-mode) in registers: read through the batch pointers they cost a global round trip per SQP
    // round (5.5 k of a round's 80 k cycles: the loads sit behind lane-dependent addresses and are not hoisted out of the loop)
    double x0r[NX], u0r[NU];

    MPCRL_DI SmallSolver(const SmallSpec &sp_, int k_, int lpi_, int base_)
        : sp(sp_), N(sp_.N), lpi(lpi_), k(k_), base(base_), blkidx(base_ / lpi_), term(k_ == sp_.N), first(k_ == 0) {}

    // ---- static problem data of this stage -------------------------------------------------------
    // bounds of THIS lane's stage in registers (init_bounds, once per launch): read out of the problem descriptor at every use they
    // cost a three-way select on the stage kind each time and keep ~40 scalar registers busy for the whole kernel
    double lbr[NW], ubr[NW];
    unsigned hasm = 0;   // bit 2 i + sd: the bound row (i, sd) exists
    MPCRL_DI void init_bounds() {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            double l, u_;
            if (first)
                l = (i < NU && !qmode) ? sp.lb0[i < NU ? i : 0] : -1e30, u_ = (i < NU && !qmode) ? sp.ub0[i < NU ? i : 0] : 1e30;
            else if (term)
                l = i >= NU ? sp.lbe[i >= NU ? i - NU : 0] : -1e30, u_ = i >= NU ? sp.ube[i >= NU ? i - NU : 0] : 1e30;
            else
                l = sp.lb[i], u_ = sp.ub[i];
            lbr[i] = l, ubr[i] = u_;
            hasm |= (l > -NO_BOUND ? 1u : 0u) << (2 * i) | (u_ < NO_BOUND ? 1u : 0u) << (2 * i + 1);
        }
    }
    MPCRL_DI double lbv(int i) const { return lbr[i]; }
    MPCRL_DI double ubv(int i) const { return ubr[i]; }
    MPCRL_DI bool has(int sd, int i) const { return (hasm >> (2 * i + sd)) & 1u; }
    // (M::soft_coord: which coordinates of v = [u; x] the model's OCP can have L1-soft bounds on — a compile-time fact, so the
    // slack state of the other coordinates never occupies registers; mpcrl_create rejects a spec that asks for more)
    MPCRL_DI bool softc(int i) const { return SOFT && M::soft_coord(i) && !first && !term && sp.soft[i] != 0; }
    MPCRL_DI double zw(int sd, int i) const { return (sd ? sp.zu[i] : sp.zl[i]) * sp.dT * pow(sp.gamma, (double)k); }
    MPCRL_DI bool fixed(int i) const { return first && (i >= NU || qmode); }
    MPCRL_DI double vc(int i) const { return i < NU ? u[i < NU ? i : 0] : x[i >= NU ? i - NU : 0]; }
    MPCRL_DI double dvc(const double *ax, const double *au, int i) const {
        return i < NU ? (term ? 0.0 : au[i < NU ? i : 0]) : ax[i >= NU ? i - NU : 0];
    }
    // A_k of the matrix-layout path (NX = 4, NU = 1) lives in the stage's LDS slot only — the linearisation writes it there, the
    // sweeps and the few stage-lane uses (residuals of a QP's first iteration, NLP stationarity) read it there: 16 doubles per lane
    // that the register allocator no longer carries through the whole interior-point loop
    MPCRL_DI double Aget(int i) const {
        if constexpr (NX == 4 && NU == 1) return ms[(threadIdx.x & 63) * 66 + 2 * i]; else return A[i];
    }
    MPCRL_DI double Bget(int i) const { return Bm[i]; }
    MPCRL_DI void Aset(int i, double v) {
        if constexpr (NX == 4 && NU == 1) ms[(threadIdx.x & 63) * 66 + 2 * i] = v; else A[i] = v;
    }
    MPCRL_DI void Bset(int i, double v) { Bm[i] = v; }
    MPCRL_DI double BA(int m, int j) const { return j < NU ? Bget(m * NU + (j < NU ? j : 0)) : Aget(m * NX + (j >= NU ? j - NU : 0)); }
    // slack(v) of the bound row on side sd, coordinate i, at value v
    MPCRL_DI double bslack(int sd, int i, double v) const {
        const double sv = SOFT && softc(i) ? s[sd][ss(i)] : 0.0;
        return sd ? ubv(i) - v + sv : v + sv - lbv(i);
    }

    // ---- linearise the dynamics leaving this stage and the stage cost; returns c_k * l_k ---------
    // WITH_H (sensitivity pass): the same evaluation with the full second-order jet, so that the exact-Hessian term
    // sum_m nu_{k+1,m} hess F_m along the non-trivial coordinates (hdd, packed lower triangle) comes out of the evaluation that
    // yields A, B — the pass used to run the map once with Jet1 for [B A] and once more with JetH for the Hessian.
    template <bool WITH_H = false>
    MPCRL_DI double linearize(const double *xnext, const double *nu_next = nullptr, double *hdd = nullptr) {
        mx_dyn_dirty = true;
        {
            // The terminal lane has no dynamics (A = B = 0, r = 0).  It runs the same code and the results are SELECTED: with an
            // if/else the optimiser sinks the two branches' stores into one block through pointer phis, which keeps part of A, B,
            // r in scratch memory for the whole kernel (a global-memory round trip at every use).
            constexpr int ND = M::NLD;   // jet directions: the coordinates of v = [u; x] whose columns are not known in closed form
            using J = std::conditional_t<WITH_H, JetH<ND>, Jet1<ND>>;
            J jx[NX], ju[NU], jt[NTD], jn[NX];
#pragma unroll
            for (int i = 0; i < NU; ++i) ju[i] = J(u[i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) jx[i] = J(x[i]);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int c = M::lin_coord(d);
                if constexpr (WITH_H) {
                    if (c < NU) ju[c < NU ? c : 0].g[d] = 1.0; else jx[c >= NU ? c - NU : 0].g[d] = 1.0;
                } else {
                    if (c < NU) ju[c < NU ? c : 0].d[d] = 1.0; else jx[c >= NU ? c - NU : 0].d[d] = 1.0;
                }
            }
#pragma unroll
            for (int i = 0; i < NTD; ++i) jt[i] = J(thd[i]);
            disc_map<M, J>(jx, ju, jt, jn, sp.h, sp.rk_steps);
            M::lin_trivial(sp.h * sp.rk_steps, [&](int i, int j, double v) { Aset(i * NX + j, term ? 0.0 : v); });
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                r[i] = term ? 0.0 : jn[i].v - xnext[i];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const int c = M::lin_coord(d);
                    double der;
                    if constexpr (WITH_H) der = jn[i].g[d]; else der = jn[i].d[d];
                    if (c < NU)
                        Bset(i * NU + (c < NU ? c : 0), term ? 0.0 : der);
                    else
                        Aset(i * NX + (c >= NU ? c - NU : 0), term ? 0.0 : der);
                }
            }
            if constexpr (WITH_H) {
#pragma unroll
                for (int e = 0; e < ND * (ND + 1) / 2; ++e) {
                    double acc = 0.0;
#pragma unroll
                    for (int m = 0; m < NX; ++m) acc = fma(nu_next[m], jn[m].h[e], acc);
                    hdd[e] = term ? 0.0 : acc;
                }
            }
        }
        double val = M::cost_grad(term, k, x, u, sp, thc.ptr(), Hc, ctab + NHT, q);
#pragma unroll
        for (int i = 0; i < NW; ++i) q[i] *= ck;
        val *= ck;
        if constexpr (SOFT) {
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if (softc(i)) val += zw(0, i) * s[0][ss(i)] + zw(1, i) * s[1][ss(i)];
        }
        return val;
    }

    // (G' nu) contribution to the stationarity row of coordinate i: [B A]' nu_{k+1} - [0; nu_k]
    MPCRL_DI double GTnu(const double *nu_next, const double *nu_own, int i) const {
        double a = 0.0;
        if (!term) {
#pragma unroll
            for (int m = 0; m < NX; ++m) a = fma(BA(m, i), nu_next[m], a);
        }
        if (i >= NU && !first) a -= nu_own[i >= NU ? i - NU : 0];
        return a;
    }

    // ---- NLP residuals (stationarity, equality, inequality, complementarity), local maxima -------
    MPCRL_DI void nlp_res_local(const double *nu_next, const double *x0, const double *u0f, double *res) const {
        double rs = 0, re = 0, ri = 0, rc = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (term && i < NU) continue;
            if (!fixed(i)) {
                double g = q[i] + GTnu(nu_next, nu_, i);
                if (has(0, i)) g -= lam[0][i];
                if (has(1, i)) g += lam[1][i];
                rs = fmax(rs, fabs(g));
            }
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
                if (has(sd, i)) {
                    const double h = -bslack(sd, i, vc(i));
                    ri = fmax(ri, h);
                    rc = fmax(rc, fabs(lam[sd][i] * h));
                    if constexpr (SOFT) {
                        if (softc(i)) {
                            ri = fmax(ri, -s[sd][ss(i)]);
                            rc = fmax(rc, fabs(lams[sd][ss(i)] * s[sd][ss(i)]));
                            rs = fmax(rs, fabs(zw(sd, i) - lam[sd][i] - lams[sd][ss(i)]));
                        }
                    }
                }
        }
        if (!term) {
#pragma unroll
            for (int i = 0; i < NX; ++i) re = fmax(re, fabs(r[i]));
        }
        if (first) {
#pragma unroll
            for (int i = 0; i < NX; ++i) re = fmax(re, fabs(x[i] - x0[i]));
            if (qmode) {
#pragma unroll
                for (int i = 0; i < NU; ++i) re = fmax(re, fabs(u[i] - u0f[i]));
            }
        }
        res[0] = rs, res[1] = re, res[2] = ri, res[3] = rc;
    }

    // ---- one backward Riccati stage: (Pn, pn) of stage k+1  ->  K, Li, kff, P, p of this stage ----
    // FACTOR = false re-uses K, Li and P (vector-only sweep for the corrector / extra right-hand sides).
    // Hs(i,j): UNSCALED stage Hessian accessor, multiplied by hscale where it is used (keeping the product out of registers:
    // a per-lane scale times 15-25 kernel-argument constants would otherwise be hoisted and held live across the whole loop);
    // Dg: barrier diagonal, g: modified gradient, bb: dynamics offset.
    // VEC = false: the matrix part only (P_k, K_k, 1/R_k) — tried with the vector recursion as a scan after the factor sweep: slower
    // (3.93 vs 4.51 M solves/s: the scan costs more than the ~20 instructions it takes out of a factor step)
    template <bool FACTOR, class HF, bool VEC = true>
    MPCRL_DI bool riccati_stage(const double *Pn, const double *pn, HF Hs, const double *g, const double *bb) {
        bool ok = true;
        double cc[NX], mv[NW];
        if (term) {
            if constexpr (FACTOR) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) P[sym(i, j)] = fma(hscale, Hs(NU + i, NU + j), i == j ? Dg[NU + i] : 0.0);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) p[i] = g[NU + i];
            return true;
        }
        if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = pn[i];
            if constexpr (FACTOR) {   // vector-only sweeps receive p_{k+1} + P_{k+1} bb already summed by the lane that owns P_{k+1}
#pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(Pn[sym(i, j)], bb[j], a);
            }
            cc[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            double a = g[i];
#pragma unroll
            for (int m = 0; m < NX; ++m) a = fma(BA(m, i), cc[m], a);
            mv[i] = a;
        }
        }
        double Mm[NW * (NW + 1) / 2];
        if constexpr (FACTOR) {
            double T[NX * NW];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    double a = 0.0;
#pragma unroll
                    for (int m = 0; m < NX; ++m) a = fma(Pn[sym(i, m)], BA(m, j), a);
                    T[i * NW + j] = a;
                }
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double a = fma(hscale, Hs(i, j), i == j ? Dg[i] : 0.0);
#pragma unroll
                    for (int m = 0; m < NX; ++m) a = fma(BA(m, i), T[m * NW + j], a);
                    Mm[sym(i, j)] = a;
                }
            if (first && qmode) {
#pragma unroll
                for (int i = 0; i < NU * NX; ++i) K[i] = 0.0;
#pragma unroll
                for (int i = 0; i < NLK; ++i) Li[i] = 0.0;
            } else {
                // Cholesky R = L L' of the control block; Li holds L with the diagonal inverted
                if constexpr (NU == 1) {
                    ok = ok && (Mm[0] > 0.0);
                    Li[0] = fast_rcp(Mm[0]);   // scalar pivot: keep 1/R itself, no square root
                } else
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        double a = Mm[sym(i, j)];
#pragma unroll
                        for (int m = 0; m < j; ++m) a -= Li[sym(i, m)] * Li[sym(j, m)];
                        if (i == j) {
                            ok = ok && (a > 0.0);
                            Li[sym(i, i)] = 1.0 / sqrt(a);
                        } else
                            Li[sym(i, j)] = a * Li[sym(j, j)];
                    }
                if constexpr (NU == 1) {
#pragma unroll
                    for (int j = 0; j < NX; ++j) K[j] = Mm[sym(NU + j, 0)] * Li[0];
                } else
#pragma unroll
                for (int j = 0; j < NX; ++j) {   // K = R^{-1} S, S(i, j) = Mm(NU + j, i)
                    double y[NU];
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        double a = Mm[sym(NU + j, i)];
#pragma unroll
                        for (int m = 0; m < i; ++m) a -= Li[sym(i, m)] * y[m];
                        y[i] = a * Li[sym(i, i)];
                    }
#pragma unroll
                    for (int i = NU - 1; i >= 0; --i) {
                        double a = y[i];
#pragma unroll
                        for (int m = i + 1; m < NU; ++m) a -= Li[sym(m, i)] * K[m * NX + j];
                        K[i * NX + j] = a * Li[sym(i, i)];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double a = Mm[sym(NU + i, NU + j)];
#pragma unroll
                    for (int m = 0; m < NU; ++m) a -= Mm[sym(NU + i, m)] * K[m * NX + j];
                    P[sym(i, j)] = a;
                }
        }
        if constexpr (!VEC) return ok;
        if (first && qmode) {
#pragma unroll
            for (int i = 0; i < NU; ++i) kff[i] = 0.0;
        } else if constexpr (NU == 1) {
            kff[0] = mv[0] * Li[0];
        } else {
            double y[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = mv[i];
#pragma unroll
                for (int m = 0; m < i; ++m) a -= Li[sym(i, m)] * y[m];
                y[i] = a * Li[sym(i, i)];
            }
#pragma unroll
            for (int i = NU - 1; i >= 0; --i) {
                double a = y[i];
#pragma unroll
                for (int m = i + 1; m < NU; ++m) a -= Li[sym(m, i)] * kff[m];
                kff[i] = a * Li[sym(i, i)];
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = mv[NU + i];
#pragma unroll
            for (int m = 0; m < NU; ++m) a -= K[m * NX + i] * mv[m];
            p[i] = a;
        }
        return ok;
    }

    // =====================================================================================================================
    // ALL Riccati sweeps in MATRIX layout on the matrix cores (NX = 4, NU = 1: v_mfma_f64_4x4x4_4b_f64).
    // The stage-per-lane layout leaves the serial sweeps with one useful lane per instance and step (a factor step as ~300 wave-wide
    // VALU instructions, a vector step as 45-60).  Here the wave switches layout through LDS: each stage lane publishes its
    // linearisation to a slot, then the 64 lanes act as 4 blocks x (4 x 4) matrix elements — block = instance of the wave,
    // lane = 16 r + 4 blk + c holds element (r, c) — and one MFMA is a 4x4x4 product for all blocks at once.  Measured operand
    // layout (profiles/microbench/mfma_f64_4x4x4_probe.hip): D(i,j) at lane 16 i + 4 blk + j, B-operand(k,j) at 16 k + 4 blk + j,
    // A-operand(i,k) at 16 k + 4 blk + i — so a matrix held in the D layout is its own B operand and, as A operand, its
    // TRANSPOSE:  mfma(X, Y, C) = X' Y + C.  Row r of a block is one quad (4 consecutive lanes): every broadcast the recursion
    // needs is a quad_perm DPP; nothing crosses DPP rows.
    //
    // Factor step (P = P_{k+1}, W = [B | bb | bb | 0], pc = [0 | p_{k+1} | 0 | 0], BB = [B B B B]):
    //   X1 = mfma(P, A)            = P A                      Y  = mfma(P, W, pc) = [P B | p + P bb | P bb | 0]
    //   Q  = mfma(A, X1, Hxx + D)  = A' P A + Hxx + D_x       Zc = mfma(A, Y, [Hxu | g_x | 0 | 0]) = [S | mv_x | A' P bb | 0]  (columns)
    //   Zr = mfma(Y, A, [Hux; 0])  : row 0 = S'               Zb = mfma(BB, Y, [Huu + D_u | g_u | 0 | 0]): every row = [R, mv_u, beta, 0]
    //   K = S / R, kff = mv_u / R,  t = Zc - K Zb (own column): p_k in column 1, q_k = Acl_k' P_{k+1} bb_k in column 2,
    //   P = Q - K S' (one fma per element: S' arrives broadcast over the rows from an MFMA on the column-0 broadcast of Y),
    //   d_k = bb - B kff.  7 MFMAs + ~50 VALU per step instead of ~300 VALU instructions.
    //
    // Vector sweeps (round 3): on the closed-loop matrices Acl_k = A_k - B_k K_k (one fma per element out of A, B, K in the slot,
    // off the chain) both vector recursions are AFFINE CHAINS  v <- Acl v + c  on the matrix cores, one dependent MFMA per stage
    // (mx_chain):
    //   forward   Dx_{k+1} = Acl_k Dx_k + d_k,                 d_k = bb_k - B_k kff_k
    //   backward  p_k      = Acl_k' p_{k+1} + c_k,             c_k = q_k + g_x - K_k' g_u     (q_k does not depend on the right-hand side)
    // and everything off the chain is stage-parallel in the stage lanes: Du_k = -K_k Dx_k - kff_k, the corrector's
    // kff_k = (g_u + beta_k + B_k' p_{k+1}) / R_k, Dnu_k = P_k Dx_k + p_k.  A chain step is ~8 instructions against 45 (forward) / 60
    // (backward) of the stage-layout sweeps they replace, and the stage lanes no longer hold K, kff, 1/R, P, p (20 doubles).
    static constexpr bool MX = (NX == 4 && NU == 1);
    // Slot of one stage (doubles; 16-byte aligned pairs so that one ds_read_b128 brings two operands — LDS instructions, not
    // bytes, are what the sweeps wait for).  Everything is (re)published by the stage lane before every factor sweep:
    //   [0,32)   (A(r,c), Hxx(r,c) + D_x) pairs at 2 (4 r + c); A is written by the linearisation and stays, the factor sweep
    //            overwrites the second member with P_k(r,c)
    //   [32,48)  per row r: (B[r], Hxu[r]) at 32 + 4 r — the factor sweep / the backward chain leave p_k[r] in the second member;
    //            (bb[r], g_x[r]) at 34 + 4 r — second member: d_k[r] after the factor sweep, then c_k[r], then the corrector's d_k[r]
    //   [48,56)  per column c: ([Huu + D_u | g_u | 0 | 0][c], Hxu[c]) at 48 + 2 c — afterwards (q_k[c], Dx_k[c])
    //   56 K[4], 60 1/R (0 for a pinned u_0), 61 kff, 62 beta = B' P_{k+1} bb
    // After the 64 slots: one ok flag per block, and a write-only dump pair for lanes with nothing to store.
    static constexpr int mxAH = 0, mxCol = 32, mxRow = 48, mxK = 56, mxMisc = 60,
                         MSLOT = 66,   // 64 slots = 33 KB: four single-wave workgroups fit one CU's LDS, with room for a parked instance;
                                       // 66 doubles = 132 words = 4 mod 64 banks: the stage lanes' ds_*_b128 are conflict-free
                         MX_SLOTS = MPCRL_MX_SLOTS, mxFlag = MX_SLOTS * MSLOT, mxDump = mxFlag + 4, MX_LDS = mxDump + 2;
    double *ms = nullptr;   // LDS, 64 slots of MSLOT doubles (one per stage lane)
    // (MPCRL_MX_SLOTS < 64, an experiment build: the lanes past the last instance share the last slot as a dump)
    MPCRL_DI static int my_slot() {
        const int l = threadIdx.x & 63;
        return MX_SLOTS == 64 ? l : (l < MX_SLOTS - 1 ? l : MX_SLOTS - 1);
    }
    // LDS cost table, one per instance of the wavefront: for each stage kind (0 = stage 0, 1 = interior, 2 = terminal) the packed
    // lower triangle of the UNSCALED stage-cost Hessian and the reference point of the residual (cartpole: W_0 / W / W_e and
    // yref_0 / yref / yref_e out of the instance's parameter vector, so that set_parameter / cost_set reach the solve as they do
    // in the reference, mpc.py:233-257).  A lane keeps its own stage's Hessian set in registers (Hc); a lane-dependent offset into
    // kernel arguments instead would make the compiler copy them to scratch and index that (a global-memory round trip per use).
    static constexpr int NHT = NW * (NW + 1) / 2, CSET = NHT + NW, CTAB = 3 * CSET;
    const double *ctab = nullptr;   // this lane's set: [NHT Hessian | NW reference point]
    MPCRL_DI int stage_kind() const { return first ? 0 : (term ? 2 : 1); }
    // tab: LDS, CTAB doubles per instance slot of the wavefront; th: the instance's full parameter vector
    MPCRL_DI void fill_cost_table(double *tab, int slot_, bool owner, const double *th) {
        double *mine = tab + slot_ * CTAB;
        if (first && owner) {   // lanes past the last instance slot of the wavefront shadow slot 0 and must not write its table
            for (int kind = 0; kind < 3; ++kind) {
                double *set = mine + kind * CSET;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) set[sym(i, j)] = M::hess(kind, i, j, sp, th);
                    set[NHT + i] = M::yref(kind, i, th);
                }
            }
        }
        ctab = mine + stage_kind() * CSET;
    }
    double Hc[NHT];   // this lane's Hessian set, in registers
    MPCRL_DI double hess_of_stage(int i, int j) const { return Hc[i >= j ? sym(i, j) : sym(j, i)]; }
    MPCRL_DI void load_hc() {
#pragma unroll
        for (int e = 0; e < NHT; ++e) Hc[e] = ctab[e];
    }
    typedef double mx_d2 __attribute__((ext_vector_type(2)));

    MPCRL_DI static void wave_lds_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    template <int CSEL>
    MPCRL_DI static double quad_bcast(double v) {
        constexpr int ctrl = CSEL | (CSEL << 2) | (CSEL << 4) | (CSEL << 6);
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    MPCRL_DI static double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
    MPCRL_DI static mx_d2 lds_pair(const double *q) { return *(const mx_d2 *)__builtin_assume_aligned(q, 16); }
    MPCRL_DI static void lds_pair_store(double *q, double a, double b) {
        mx_d2 v;
        v.x = a, v.y = b;
        *(mx_d2 *)__builtin_assume_aligned(q, 16) = v;
    }

    // stage lane -> slot: everything of the stage but A_k (written by the linearisation, never overwritten); the sweeps overwrite most of it
    template <class HF>
    MPCRL_DI void mx_publish(HF Hs, const double *g, const double *bb) {
        double *sl = ms + my_slot() * MSLOT;
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j)
                sl[mxAH + 2 * (4 * i + j) + 1] = fma(hscale, Hs(NU + (i > j ? i : j), NU + (i > j ? j : i)), i == j ? Dg[NU + i] : 0.0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double hxu = hscale * Hs(NU + i, 0);
            lds_pair_store(sl + mxCol + 4 * i, Bm[i], hxu);
            lds_pair_store(sl + mxCol + 4 * i + 2, bb[i], g[NU + i]);
            lds_pair_store(sl + mxRow + 2 * i, i == 0 ? fma(hscale, Hs(0, 0), Dg[0]) : (i == 1 ? g[0] : 0.0), hxu);
        }
    }
    // the factor sweep, all lanes in matrix layout.  WANT_P: also leave p_k of this right-hand side in the slots (adjoint solves
    // of the sensitivity pass; the interior-point predictor never reads it).
    // The rank-one update P = Q - K S' is ONE fma per element: S' arrives broadcast over the rows from an MFMA on the column-0
    // broadcast of Y, so neither a seventh MFMA nor its operand selects sit on the chain.
    struct MxOps {
        mx_d2 ah, cp, rp;
        double Br;
    };
    template <bool WANT_P>
    MPCRL_DI void mx_factor() {
        const int l = threadIdx.x & 63, r = l >> 4, blk = (l >> 2) & 3, c = l & 3;
        const int ipw = min(64 / lpi, M::MAX_IPW);
        const bool live = blk < ipw;
        double *S0 = ms + (live ? blk : 0) * lpi * MSLOT;   // an idle block shadows block 0 and stores to the dump
        const int oAH = mxAH + 2 * (4 * r + c), oCol = mxCol + 4 * r + (c >= 1 ? 2 : 0), oB = mxCol + 4 * r, oRow = mxRow + 2 * c;
        // store targets: slot-relative for lanes with something to store, else the dump pair (stride 0)
        double *const dump = ms + mxDump;   // (every lane its own dump word instead of one for all: measured, no difference)
        const bool has1 = live && c < 3, has2 = live && r == 0 && c < 3, has3 = live && c == 1;
        double *const wP = live ? S0 + oAH + 1 : dump;
        double *const w1 = has1 ? S0 + (c == 0 ? mxK + r : (c == 1 ? mxCol + 4 * r + 3 : mxRow + 2 * r)) : dump;   // K[r] | d[r] | q[r]
        double *const w2 = has2 ? S0 + mxMisc + c : dump + 1;                                                      // 1/R | kff | beta
        double *const w3 = has3 ? S0 + mxCol + 4 * r + 1 : dump;                                                   // p[r]
        const int sL = live ? MSLOT : 0, s1 = has1 ? MSLOT : 0, s2 = has2 ? MSLOT : 0, s3 = has3 ? MSLOT : 0;
        double Pm, pcol;
        {
            const double *sl = S0 + N * MSLOT;
            Pm = sl[oAH + 1];
            pcol = c == 1 ? sl[oCol + 1] : 0.0;   // g_x of the terminal stage (column 1)
        }
        double okacc = 1.0;             // > 0 while every pivot was positive
        auto load = [&](MxOps &o, int kk) {
            const double *sl = S0 + (kk > 0 ? kk : 0) * MSLOT;
            o.ah = lds_pair(sl + oAH), o.cp = lds_pair(sl + oCol), o.rp = lds_pair(sl + oRow), o.Br = sl[oB];
        };
        double sV1 = 0.0, sV2 = 0.0, sV3 = 0.0;   // results of a step
        // the arithmetic of one step.  PIN: the step of stage 0 in Q-mode (u_0 pinned: K_0 = 0, kff_0 = 0, P_0 = Q)
        auto head = [&](const MxOps &o, double &Y, double &X1) {
            Y = mfma4(Pm, o.cp.x, pcol);          // column 3 of W is never read: bb there as well (no select)
            X1 = mfma4(Pm, o.ah.x, 0.0);
        };
        auto tail = [&](const MxOps &o, double Y, double X1, bool pin) {
#if MPCRL_V_ASMSEL
            const double Am = o.ah.x, CZc = lane_select(LANES_C01, o.cp.y, 0.0);
#else
            const double Am = o.ah.x, CZc = c < 2 ? o.cp.y : 0.0;
#endif
            const double Yb = quad_bcast<0>(Y);
            const double Zc = mfma4(Am, Y, CZc);
            const double Zb = mfma4(o.Br, Y, o.rp.x);
            const double Qt = mfma4(Am, X1, o.ah.y);
            const double Zr = mfma4(Yb, Am, o.rp.y);            // every row = S'
            const double R = quad_bcast<0>(Zb), mvu = quad_bcast<1>(Zb), Sr = quad_bcast<0>(Zc);
            okacc = (R > 0.0 || pin) ? okacc : 0.0;
            const double Rinv = pin ? 0.0 : fast_rcp(R);
            const double Kr = Sr * Rinv;
            Pm = fma(-Kr, Zr, Qt);                              // P_k = Q - K S'
            const double t = fma(-Kr, Zb, Zc);                  // own column: p_k (c = 1), q_k (c = 2)
#if MPCRL_V_ASMSEL
            pcol = lane_select(LANES_C1, t, 0.0);
#else
            pcol = c == 1 ? t : 0.0;
#endif
            const double kf = mvu * Rinv;
            const double dk = fma(-o.Br, kf, o.cp.x);           // c = 1: bb[r] - B[r] kff
#if MPCRL_V_ASMSEL
            sV1 = lane_select(LANES_C0, Kr, lane_select(LANES_C1, dk, t));       // K[r] | d[r] | q[r]
            sV2 = lane_select(LANES_C0, Rinv, lane_select(LANES_C1, kf, Zb));    // 1/R | kff | beta
#else
            sV1 = c == 0 ? Kr : (c == 1 ? dk : t);              // K[r] | d[r] | q[r]
            sV2 = c == 0 ? Rinv : (c == 1 ? kf : Zb);           // 1/R | kff | beta
#endif
            sV3 = t;
        };
        auto store = [&](int kk) {
            wP[kk * sL] = Pm;
            w1[kk * s1] = sV1;
            w2[kk * s2] = sV2;
            if constexpr (WANT_P) w3[kk * s3] = sV3;
        };
        // the operands of a stage do not depend on the recursion: those of stage kk - 1 are fetched while stage kk is computed
        MxOps on;
        load(on, N - 1);
        for (int kk = N - 1; kk >= 0; --kk) {
            const MxOps o = on;
            load(on, kk - 1);
            double Y, X1;
            head(o, Y, X1);
            tail(o, Y, X1, kk == 0 && qmode);
            store(kk);
        }
        if (live && r == 0 && c == 0) ms[mxFlag + blk] = okacc;
    }
    // One affine chain over the horizon on the matrix cores, all four columns of a block carrying the same vector:
    //   FWD   v_{k+1} = Acl_k v_k + d_k   (k = 0 .. N-1, v_0 = 0),    d_k at [35 + 4 r] of slot k, v_{k+1} -> [49 + 2 r] of slot k+1
    //   !FWD  v_k = Acl_k' v_{k+1} + c_k  (k = N-1 .. 0, v_N = c_N),  c_k at [35 + 4 r] of slot k, v_k     -> [33 + 4 r] of slot k
    // mfma(X, V, C) = X'V + C: the forward chain needs X(r,c) = Acl(c,r) = A(c,r) - B[c] K[r], the backward chain X(r,c) = Acl(r,c) =
    // A(r,c) - B[r] K[c]; the operands of a step are fetched DEPTH steps ahead, so forming X is off the chain.
    template <bool FWD>
    MPCRL_DI void mx_chain() {
        const int l = threadIdx.x & 63, r = l >> 4, blk = (l >> 2) & 3, c = l & 3;
        const int ipw = min(64 / lpi, M::MAX_IPW);
        const bool live = blk < ipw;
        const double *S0 = ms + (live ? blk : 0) * lpi * MSLOT;
        const int oA = mxAH + (FWD ? 2 * (4 * c + r) : 2 * (4 * r + c)), oBx = mxCol + 4 * (FWD ? c : r), oKx = mxK + (FWD ? r : c),
                  oC = mxCol + 4 * r + 3;
        const bool wr = live && c == 0;
        double *const wO = wr ? ms + (S0 - ms) + (FWD ? mxRow + 2 * r + 1 : mxCol + 4 * r + 1) : ms + mxDump;
        const int sO = wr ? MSLOT : 0;
        double V = 0.0;
        if constexpr (!FWD) {
            V = S0[N * MSLOT + oC];
            wO[N * sO] = V;   // p_N
        }
        // operands DEPTH steps ahead in a ring of register sets: a load issued in one step is first waited for DEPTH steps later, so
        // the LDS round trip (longer than the MFMA's dependent latency) stays off the chain
#ifndef MPCRL_CHAIN_DEPTH
#define MPCRL_CHAIN_DEPTH 3   // measured: 3 beats 2 and 4 (4 costs registers the kernel does not have)
#endif
        constexpr int DEPTH = MPCRL_CHAIN_DEPTH;
        auto slot_of = [&](int s_) { const int sc = s_ < N ? s_ : N - 1; return FWD ? sc : N - 1 - sc; };   // stage of chain step s_ (clamped)
        double Ar[DEPTH], Br_[DEPTH], Kr_[DEPTH], Cr[DEPTH];
        auto fetch = [&](int j, int s_) {
            const double *sl = S0 + slot_of(s_) * MSLOT;
            Ar[j] = sl[oA], Br_[j] = sl[oBx], Kr_[j] = sl[oKx], Cr[j] = sl[oC];
        };
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) fetch(j, j);
        int s_ = 0;
        for (; s_ + DEPTH <= N; s_ += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                V = mfma4(fma(-Br_[j], Kr_[j], Ar[j]), V, Cr[j]);
                wO[(FWD ? slot_of(s_ + j) + 1 : slot_of(s_ + j)) * sO] = V;
                fetch(j, s_ + j + DEPTH);
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j)
            if (s_ + j < N) {
                V = mfma4(fma(-Br_[j], Kr_[j], Ar[j]), V, Cr[j]);
                wO[(FWD ? slot_of(s_ + j) + 1 : slot_of(s_ + j)) * sO] = V;
            }
    }
    // Predictor / adjoint solve: publish, factor, forward chain; the stage lane gets Dx, Du (and with WANT_P the factors needed for
    // Dnu, see mx_dnu).  Returns false where the factorisation met a non-positive pivot.
    template <bool WANT_P, class HF>
    MPCRL_DI bool mx_pred(HF Hs, const double *g, const double *bb) {
        mx_publish(Hs, g, bb);
        wave_lds_sync();
        PHW(12);
        mx_factor<WANT_P>();
        wave_lds_sync();
        PHW(13);
        mx_chain<true>();
        wave_lds_sync();
        PHW(3);
        const double *sl = ms + my_slot() * MSLOT;
        const mx_d2 k01 = lds_pair(sl + mxK), k23 = lds_pair(sl + mxK + 2), rk = lds_pair(sl + mxMisc);
        const double Kl[NX] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
        for (int i = 0; i < NX; ++i) Dx[i] = first ? 0.0 : sl[mxRow + 2 * i + 1];
        double a = -rk.y;
#pragma unroll
        for (int j = 0; j < NX; ++j) a = fma(-Kl[j], Dx[j], a);
        Du[0] = term ? 0.0 : a;
        PHW(14);
        return ms[mxFlag + blkidx] != 0.0;
    }
    // Corrector: new right-hand side g (bb unchanged), same factorisation: backward chain, feed-forward terms, forward chain
    MPCRL_DI void mx_corr(const double *g, const double *bb) {
        double *sl = ms + my_slot() * MSLOT;
        const mx_d2 k01 = lds_pair(sl + mxK), k23 = lds_pair(sl + mxK + 2);
        const double Kl[NX] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double q_ = sl[mxRow + 2 * i];
            sl[mxCol + 4 * i + 3] = term ? g[NU + i] : fma(-Kl[i], g[0], q_ + g[NU + i]);   // c_k = q_k + g_x - K' g_u; c_N = g_x
        }
        wave_lds_sync();
        PHW(5);
        mx_chain<false>();
        wave_lds_sync();
        PHW(6);
        // kff_k = (g_u + beta_k + B_k' p_{k+1}) / R_k,  d_k = bb_k - B_k kff_k   (stage-parallel: p_{k+1} is read out of the next slot)
        const double *nx_ = (term || (threadIdx.x & 63) >= MX_SLOTS - 1) ? sl : sl + MSLOT;   // (the last slot has none behind it)
        const mx_d2 rk = lds_pair(sl + mxMisc);
        double mvu = g[0] + sl[mxMisc + 2];
#pragma unroll
        for (int i = 0; i < NX; ++i) mvu = fma(Bm[i], nx_[mxCol + 4 * i + 1], mvu);
        const double kffc = mvu * rk.x;
#pragma unroll
        for (int i = 0; i < NX; ++i) sl[mxCol + 4 * i + 3] = fma(-Bm[i], kffc, bb[i]);
        wave_lds_sync();
        PHW(2);
        mx_chain<true>();
        wave_lds_sync();
        PHW(7);
#pragma unroll
        for (int i = 0; i < NX; ++i) Dx[i] = first ? 0.0 : sl[mxRow + 2 * i + 1];
        double a = -kffc;
#pragma unroll
        for (int j = 0; j < NX; ++j) a = fma(-Kl[j], Dx[j], a);
        Du[0] = term ? 0.0 : a;
        mx_dnu(g, false);
        PHW(14);
    }
    // Dnu_k = P_k Dx_k + p_k (multipliers of the arriving dynamics).  own_p: the terminal lane's p_N is its own g_x (predictor /
    // adjoint right-hand side, where the slots hold p_k for k < N only); after the backward chain p_N is in the slot as well
    MPCRL_DI void mx_dnu(const double *g, bool own_p) {
        const double *sl = ms + my_slot() * MSLOT;
        double Pl[NPK];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Pl[sym(i, j)] = sl[mxAH + 2 * (4 * i + j) + 1];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = (own_p && term) ? g[NU + i] : sl[mxCol + 4 * i + 1];
#pragma unroll
            for (int j = 0; j < NX; ++j) a = fma(Pl[sym(i, j)], Dx[j], a);
            Dnu[i] = first ? 0.0 : a;
        }
    }
    bool mx_dyn_dirty = true;   // (kept for the stage-layout path's linearize; the matrix-layout path republishes everything)

    // ---- backward sweep over the horizon (serial in k; the lanes of all instances in the wave step together).
    // (P, p) of stage k+1 arrive by a one-lane shift.  For the vector-only sweeps (corrector, extra right-hand sides) the
    // product P_{k+1} bb_k is formed beforehand, stage-parallel, by the lane that owns P_{k+1}, so only NX values travel.
    template <bool FACTOR, class HF>
    MPCRL_DI bool backward(HF Hs, const double *g, const double *bb) {
        // Every lane executes every stage step (full EXEC mask).
        bool ok = true;
        double hb[NX];
        if constexpr (!FACTOR) {
            double bbp[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) bbp[i] = lane_up(bb[i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(P[sym(i, j)], bbp[j], a);
                hb[i] = a;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) hb[i] = 0.0;
        }
        // No commit/select is needed: a lane whose stage is already final recomputes it from its (final) neighbour and gets the
        // same bits again; a lane whose turn has not come yet computes throw-away values that its own turn overwrites.
        for (int kk = N; kk >= 0; --kk) {
            double pn[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) pn[i] = lane_dn(p[i] + hb[i]);
            double Pn[NPK];
            if constexpr (FACTOR) {
#pragma unroll
                for (int i = 0; i < NPK; ++i) Pn[i] = lane_dn(P[i]);
                const bool okk = riccati_stage<true>(Pn, pn, Hs, g, bb);
                ok = ok && (okk || k != kk);
            } else {
#pragma unroll
                for (int i = 0; i < NPK; ++i) Pn[i] = 0.0;
                riccati_stage<false>(Pn, pn, Hs, g, bb);
            }
        }
        return ok;
    }

    // =====================================================================================================================
    // Vector sweeps of the stage-per-lane layout as PARALLEL SCANS (round 4; the linear-system model: NX = 2, NU = 1, 41 stage lanes).
    // On stored factors both vector recursions are affine,
    //     backward  p_k      = Acl_k' (p_{k+1} + hb_{k+1}) + c_k,   c_k = g_x - K_k' g_u,   hb_{k+1} = P_{k+1} bb_k
    //     forward   Dx_{k+1} = Acl_k Dx_k + d_k,                     d_k = bb_k - B_k kff_k,  Acl_k = A_k - B_k K_k,
    // and affine maps compose associatively: (M2, v2) o (M1, v1) = (M2 M1, M2 v1 + v2).  A Hillis-Steele scan over the lanes of an
    // instance gives every stage its composed map in ceil(log2(N + 1)) = 6 steps of (6 doubles through ds_bpermute + 6 fma pairs)
    // instead of N + 1 = 41 serial steps in which ONE lane of the wavefront works (round 3: the three vector sweeps of an
    // interior-point iteration were 31 % of a wavefront's life).  Everything off the recursion (kff, Du, Dnu) is stage-local.
    // Same numbers as the serial sweeps up to the association of the products (the port runs the serial order).
    static constexpr bool SCAN = !MX && NU == 1 && MPCRL_SMALL_SCAN != 0;
    MPCRL_DI void compose_from(double (&Mm)[NX * NX], double (&vv)[NX], int s, bool up, bool valid) {
        // partner = lane -/+ s; this lane's map is applied AFTER (forward: partner covers earlier stages) / BEFORE ... in both sweeps
        // the own map is the OUTER one: new = own o partner
        double Mp[NX * NX], vp[NX];
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Mp[i] = up ? __shfl_up(Mm[i], s) : __shfl_down(Mm[i], s);
#pragma unroll
        for (int i = 0; i < NX; ++i) vp[i] = up ? __shfl_up(vv[i], s) : __shfl_down(vv[i], s);
        double Mn[NX * NX], vn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = vv[i];
#pragma unroll
            for (int m = 0; m < NX; ++m) a = fma(Mm[i * NX + m], vp[m], a);
            vn[i] = a;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double b = 0.0;
#pragma unroll
                for (int m = 0; m < NX; ++m) b = fma(Mm[i * NX + m], Mp[m * NX + j], b);
                Mn[i * NX + j] = b;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) vv[i] = valid ? vn[i] : vv[i];
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) Mm[i] = valid ? Mn[i] : Mm[i];
    }
    // vector-only backward sweep (what backward<false> computes: p_k, kff_k on the stored K, Li, P)
    MPCRL_DI void backward_scan(const double *g, const double *bb) {
        double hb[NX], hbn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) a = fma(P[sym(i, j)], lane_up(bb[j]), a);
            hb[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) hbn[i] = lane_dn(hb[i]);
        double Mm[NX * NX], vv[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double c = g[NU + i];
#pragma unroll
            for (int u_ = 0; u_ < NU; ++u_) c = fma(-K[u_ * NX + i], g[u_], c);
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                double a = Aget(m * NX + i);
#pragma unroll
                for (int u_ = 0; u_ < NU; ++u_) a = fma(-Bget(m * NU + u_), K[u_ * NX + i], a);
                Mm[i * NX + m] = term ? 0.0 : a;          // (Acl')(i, m); the terminal lane is the constant map p_N = g_x
                c = term ? c : fma(a, hbn[m], c);
            }
            vv[i] = term ? g[NU + i] : c;
        }
        for (int s = 1; s <= N; s <<= 1) compose_from(Mm, vv, s, false, k + s <= N);
#pragma unroll
        for (int i = 0; i < NX; ++i) p[i] = vv[i];
        // feed-forward: kff = (g_u + B' (p_{k+1} + hb_{k+1})) / R
        double mvu = g[0];
#pragma unroll
        for (int m = 0; m < NX; ++m) mvu = fma(Bget(m * NU), lane_dn(p[m] + hb[m]), mvu);
        if (!term) kff[0] = (first && qmode) ? 0.0 : mvu * Li[0];
    }
    // forward sweep (what forward() computes: Dx, Du, Dnu)
    MPCRL_DI void forward_scan(const double *bb) {
        double Mm[NX * NX], vv[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double d = bb[i];
#pragma unroll
            for (int u_ = 0; u_ < NU; ++u_) d = fma(-Bget(i * NU + u_), kff[u_], d);
            vv[i] = term ? 0.0 : d;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double a = Aget(i * NX + j);
#pragma unroll
                for (int u_ = 0; u_ < NU; ++u_) a = fma(-Bget(i * NU + u_), K[u_ * NX + j], a);
                Mm[i * NX + j] = term ? (i == j ? 1.0 : 0.0) : a;
            }
        }
        for (int s = 1; s <= N; s <<= 1) compose_from(Mm, vv, s, true, k - s >= 0);
        // lane k now holds Dx_{k+1} (its composed map applied to Dx_0 = 0)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double xin = lane_up(vv[i]);
            Dx[i] = first ? 0.0 : xin;
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            double a = -kff[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) a = fma(-K[i * NX + j], Dx[j], a);
            Du[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = p[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) a = fma(P[sym(i, j)], Dx[j], a);
            Dnu[i] = first ? 0.0 : a;
        }
    }

    // ---- forward sweep: Newton step (Dx, Du) and the multipliers Dnu of the arriving dynamics ------
    MPCRL_DI void forward(const double *bb) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Dx[i] = 0.0, Dnu[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) Du[i] = 0.0;
        double xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = 0.0;
        for (int kk = 0; kk < N; ++kk) {
            double tu[NU], tx[NX];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = -kff[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(-K[i * NX + j], Dx[j], a);
                tu[i] = a;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = bb[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(A[i * NX + j], Dx[j], a);
#pragma unroll
                for (int j = 0; j < NU; ++j) a = fma(Bm[i * NU + j], tu[j], a);
                tx[i] = a;
            }
            // as in backward(): stages that are already final (k <= kk + 1) are recomputed to the same bits, later ones are provisional
#pragma unroll
            for (int i = 0; i < NU; ++i) Du[i] = tu[i];
            double xin[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xin[i] = lane_up(tx[i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) Dx[i] = first ? 0.0 : xin[i];
        }
        // multipliers of the arriving dynamics: local to each stage once Dx is known
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double a = p[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) a = fma(P[sym(i, j)], Dx[j], a);
            Dnu[i] = first ? 0.0 : a;
        }
    }

    // ---- interior point: per-row Newton quantities ------------------------------------------------
    // complementarity target r_m of a row (affine: lam t; corrector: lam t + dlam_aff dt_aff - sigma mu)
    MPCRL_DI static double rm_(double l, double tt, double af, int pass, double smu) { return fma(l, tt, pass ? af - smu : 0.0); }

    // barrier diagonal and right-hand-side term of coordinate i (both sides), Newton system of DESIGN.md §IPM
    MPCRL_DI void barrier_terms(int i, double v, int pass, double smu, double &dg, double &er) const {
        dg = 0.0, er = 0.0;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            if (!has(sd, i)) continue;
            const double sg = sd ? -1.0 : 1.0;
            const double l1 = lam[sd][i], t1 = t[sd][i], it1 = fast_rcp(t1);
            const double w1 = l1 * it1;
            const double rd1 = t1 - bslack(sd, i, v);
            const double e1 = (rm_(l1, t1, aff[sd][i], pass, smu) - l1 * rd1) * it1;
            if (SOFT && softc(i)) {
                const int ii = ss(i);
                const double l2 = lams[sd][ii], t2 = ts[sd][ii], it2 = fast_rcp(t2);
                const double w2 = l2 * it2;
                const double e2 = (rm_(l2, t2, affs[sd][ii], pass, smu) - l2 * (t2 - s[sd][ii])) * it2;
                const double rgs = zw(sd, i) - l1 - l2;
                const double iw = fast_rcp(w1 + w2);
                dg += w1 * w2 * iw;
                er += sg * (e1 * w2 - w1 * (rgs + e2)) * iw;
            } else {
                dg += w1;
                er += sg * e1;
            }
        }
    }
    // steps of the rows of coordinate i, side sd, for a primal step dv of the coordinate.  rat = the largest of -dlam/lam and
    // -dt/t over these rows: the step to the boundary is 1 / max(rat), so no division is needed per row.
    MPCRL_DI void row_steps(int i, int sd, double v, double dv, int pass, double smu, double &dt1, double &dl1, double &dt2,
                            double &dl2, double &dss, double &rat) const {
        const double sg = sd ? -1.0 : 1.0;
        const double l1 = lam[sd][i], t1 = t[sd][i], it1 = fast_rcp(t1);
        const double rd1 = t1 - bslack(sd, i, v);
        const double rm1 = rm_(l1, t1, aff[sd][i], pass, smu);
        dss = 0.0, dt2 = 0.0, dl2 = 0.0, rat = 0.0;
        if (SOFT && softc(i)) {
            const int ii = ss(i);
            const double l2 = lams[sd][ii], t2 = ts[sd][ii], it2 = fast_rcp(t2);
            const double w1 = l1 * it1, w2 = l2 * it2;
            const double rd2 = t2 - s[sd][ii];
            const double rm2 = rm_(l2, t2, affs[sd][ii], pass, smu);
            const double e1 = (rm1 - l1 * rd1) * it1, e2 = (rm2 - l2 * rd2) * it2;
            const double rgs = zw(sd, i) - l1 - l2;
            dss = -(rgs + e1 + e2 + sg * w1 * dv) * fast_rcp(w1 + w2);
            dt2 = -rd2 + dss;
            dl2 = (-rm2 - l2 * dt2) * it2;
            // (predictor: r_m = lam t, so -dlam / lam = (t + dt) / t — no reciprocal of the multiplier)
            rat = fmax(pass ? -dl2 * fast_rcp(l2) : (t2 + dt2) * it2, -dt2 * it2);
        }
        dt1 = -rd1 + sg * dv + dss;
        dl1 = (-rm1 - l1 * dt1) * it1;
        rat = fmax(rat, fmax(pass ? -dl1 * fast_rcp(l1) : (t1 + dt1) * it1, -dt1 * it1));
    }

    // ---- Mehrotra predictor-corrector on the QP of the current linearisation -----------------------
    // act: this instance takes part.  Returns true when converged; n_it counts iterations of this instance.
    // warm_mu > 0: start from the rows and multipliers of the previous QP, every complementarity product raised to >= warm_mu.
    MPCRL_DI bool qp_solve(bool act, const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu) {
        auto Hs = [&](int i, int j) { return hess_of_stage(i, j); };
        constexpr bool SKIPC = MX && M::SKIP_CORRECTOR;
        hscale = ck;
        const bool warm = warm_mu > 0.0;
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) dx[i] = first ? x0[i] - x[i] : 0.0, nuq[i] = warm ? nu_[i] : 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) du[i] = (first && qmode) ? u0f[i] - u[i] : 0.0;
        }
        auto recentre = [&](double &l, double &tt) {
            if (l * tt < warm_mu) {
                if (l >= tt)
                    tt = warm_mu * fast_rcp(l);
                else
                    l = warm_mu * fast_rcp(tt);
            }
        };
        double cnt = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (term && i < NU) continue;
            const double v = vc(i) + dvc(dx, du, i);
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                if (!has(sd, i)) continue;
                cnt += 1.0;
                if (SOFT && softc(i)) {
                    cnt += 1.0;
                    const int ii = ss(i);
                    if (act) {
                        if (warm) {
                            double l = lams[sd][ii], tt = fmax(s[sd][ii], ts[sd][ii]);
                            recentre(l, tt);
                            lams[sd][ii] = l, ts[sd][ii] = tt;
                        } else
                            s[sd][ii] = 0.0, ts[sd][ii] = IPM_T_MIN, lams[sd][ii] = IPM_MU0 / IPM_T_MIN;
                    }
                }
                if (act) {
                    const double sl = bslack(sd, i, v);
                    if (warm) {
                        double l = lam[sd][i], tt = fmax(sl, t[sd][i]);
                        recentre(l, tt);
                        lam[sd][i] = l, t[sd][i] = tt;
                    } else {
                        t[sd][i] = fmax(sl, IPM_T_MIN);
                        lam[sd][i] = IPM_MU0 * fast_rcp(t[sd][i]);
                    }
                }
            }
        }
        // number of bound rows of the instance: a property of the problem's structure, the same for every QP of the launch
        if (n_rows_c < 0.0) n_rows_c = seg_sum<M::SEG_SKIP>(cnt, k, lpi, base);
        const double n_rows = n_rows_c;
        PHW(11);
        bool qlive = act, ok = false;
        double rinf_c = 0.0, musum_c = 0.0;   // residual norm and sum lam t carried to the next iteration
        for (int it = 0;; ++it) {
            // ---- residuals.  Evaluated for the first iteration of a QP only: every equation but the complementarity products is LINEAR in
            // (dx, du, nuq, lam, t, s), so a step of length alpha along a direction that solves the Newton system takes all their
            // residuals (r_b, r_g, the bound rows t - slack, the soft rows) to (1 - alpha) times their value exactly, and
            // sum lam t becomes a quadratic in alpha whose coefficients ride in the reduction of the step length.  Later
            // iterations scale what they have (MPCRL_IPM_SCALE_RES; the oracle re-evaluates everything every iteration, the parity
            // tests are the check that the scaled residuals are the true ones to rounding).
            double rinf, musum;
            if (!MPCRL_IPM_SCALE_RES || it == 0) {
                double dxn[NX], nuqn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) dxn[i] = lane_dn(dx[i]), nuqn[i] = lane_dn(nuq[i]);
                double rloc = 0.0, muloc = 0.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double a = 0.0;
                    if (!term) {
                        a = r[i] - dxn[i];
#pragma unroll
                        for (int j = 0; j < NX; ++j) a = fma(Aget(i * NX + j), dx[j], a);
#pragma unroll
                        for (int j = 0; j < NU; ++j) a = fma(Bget(i * NU + j), du[j], a);
                    }
                    rb[i] = a;
                    rloc = fmax(rloc, fabs(a));
                }
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    rg[i] = 0.0;
                    if (term && i < NU) continue;
                    double a = q[i] + GTnu(nuqn, nuq, i), hdv = 0.0;
#pragma unroll
                    for (int j = 0; j < NW; ++j) hdv = fma(Hs(i, j), dvc(dx, du, j), hdv);
                    a = fma(hscale, hdv, a);
                    if (has(0, i)) a -= lam[0][i];
                    if (has(1, i)) a += lam[1][i];
                    if (fixed(i)) a = 0.0;
                    rg[i] = a;
                    rloc = fmax(rloc, fabs(a));
                    const double v = vc(i) + dvc(dx, du, i);
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        if (!has(sd, i)) continue;
                        rloc = fmax(rloc, fabs(t[sd][i] - bslack(sd, i, v)));
                        muloc = fma(lam[sd][i], t[sd][i], muloc);
                        if (SOFT && softc(i)) {
                            const int ii = ss(i);
                            rloc = fmax(rloc, fabs(ts[sd][ii] - s[sd][ii]));
                            rloc = fmax(rloc, fabs(zw(sd, i) - lam[sd][i] - lams[sd][ii]));
                            muloc = fma(lams[sd][ii], ts[sd][ii], muloc);
                        }
                    }
                }
                seg_reduce<1, 1, M::SEG_SKIP>(&rloc, &muloc, k, lpi, base);
                rinf = rloc, musum = muloc;   // musum: sum of the complementarity products
            } else
                rinf = rinf_c, musum = musum_c;
            // (scalars of the iteration by the refined hardware reciprocal, as everywhere else: their operands are positive and normal)
            const double inv_rows = n_rows > 0.0 ? fast_rcp(n_rows) : 0.0;
            const double mu = musum * inv_rows;
            if (qlive) {
                if (rinf <= tol_res && mu <= tol_mu)
                    qlive = false, ok = true;
                else if (it >= IPM_MAX_ITER || !(rinf < 1e300))
                    qlive = false;
            }
            if (!__any(qlive)) break;
            if (qlive) ++n_it;
            PHW(0);
            // ---- predictor
            double eaff[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const double v = vc(i) + dvc(dx, du, i);
                barrier_terms(i, v, 0, 0.0, Dg[i], eaff[i]);
                rt[i] = rg[i] + eaff[i];
            }
            bool okf;
            PHW(1);
            if constexpr (MX)
                okf = mx_pred<SKIPC>(Hs, rt, rb);   // (with the p_k of this right-hand side where the predictor step may be the step)
            else {
                okf = backward<true>(Hs, rt, rb);
                PHW(2);
                if constexpr (SCAN) forward_scan(rb); else forward(rb);
                PHW(3);
            }
            double okbad = okf ? 0.0 : 1.0;   // reduced together with the predictor's step length below
            double rmax = 1.0;                // rmax = 1 / (step to the boundary), at least 1
            // mu_aff(a) = sum (lam + a dlam)(t + a dt) = sum lam t + a c1 + a^2 c2: the two sums travel in the reduction tree of the step
            // length, so the predictor needs ONE cross-lane reduction and one pass over its rows (it was two of each: the rows were
            // walked again, Newton quantities and all, once the step length was known)
            double c12[2] = {0.0, 0.0};
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (term && i < NU) continue;
                const double v = vc(i) + dvc(dx, du, i), dv = dvc(Dx, Du, i);
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    if (!has(sd, i)) continue;
                    double dt1, dl1, dt2, dl2, dss, rat;
                    row_steps(i, sd, v, dv, 0, 0.0, dt1, dl1, dt2, dl2, dss, rat);
                    rmax = fmax(rmax, rat);
                    aff[sd][i] = dl1 * dt1;   // only read by the corrector (pass = 1)
                    c12[0] = fma(lam[sd][i], dt1, fma(t[sd][i], dl1, c12[0])), c12[1] = fma(dl1, dt1, c12[1]);
                    if (SOFT && softc(i)) {
                        const int ii = ss(i);
                        affs[sd][ii] = dl2 * dt2;
                        c12[0] = fma(lams[sd][ii], dt2, fma(ts[sd][ii], dl2, c12[0])), c12[1] = fma(dl2, dt2, c12[1]);
                    }
                }
            }
            {
                double two[2] = {rmax, okbad};
                seg_reduce<2, 2, M::SEG_SKIP>(two, c12, k, lpi, base);
                rmax = two[0];
                if (two[1] > 0.5) qlive = false;   // non-positive pivot: QP failure
            }
            const double a_aff = fast_rcp(rmax);
            const double mu_aff = fma(a_aff, fma(a_aff, c12[1], c12[0]), musum) * inv_rows;
            const double ratio = mu > 0.0 ? mu_aff * fast_rcp(mu) : 0.0;
            const double sig3 = ratio * ratio * ratio;
            const double smu = sig3 * mu;
            const double frac = (M::DISCRETE || M::EXACT_QP) ? IPM_FRAC : fmax(IPM_FRAC, 1.0 - mu);   // fraction to the boundary -> 1 as mu -> 0 (LQ model, exact-QP mode: fixed)
            // the step along (Dx, Du, Dnu) and the rows' directions of `pass`: rows of (i, sd) only read their own side's state, so they
            // can be advanced in place; e12: sum lam t after the step = musum + alpha e12[0] + alpha^2 e12[1]
            auto take_step = [&](int pass, double smu_, double alpha, const double *e12) {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (term && i < NU) continue;
                    const double v = vc(i) + dvc(dx, du, i), dv = dvc(Dx, Du, i);
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        if (!has(sd, i)) continue;
                        double dt1, dl1, dt2, dl2, dss, rat;
                        row_steps(i, sd, v, dv, pass, smu_, dt1, dl1, dt2, dl2, dss, rat);
                        lam[sd][i] = fma(alpha, dl1, lam[sd][i]);
                        t[sd][i] = fma(alpha, dt1, t[sd][i]);
                        if (SOFT && softc(i)) {
                            const int ii = ss(i);
                            lams[sd][ii] = fma(alpha, dl2, lams[sd][ii]);
                            ts[sd][ii] = fma(alpha, dt2, ts[sd][ii]);
                            s[sd][ii] = fma(alpha, dss, s[sd][ii]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) dx[i] = fma(alpha, Dx[i], dx[i]), nuq[i] = fma(alpha, Dnu[i], nuq[i]);
#pragma unroll
                for (int i = 0; i < NU; ++i) du[i] = fma(alpha, Du[i], du[i]);
                if (MPCRL_IPM_SCALE_RES) {
                    const double om = 1.0 - alpha;
#pragma unroll
                    for (int i = 0; i < NX; ++i) rb[i] *= om;
#pragma unroll
                    for (int i = 0; i < NW; ++i) rg[i] *= om;
                    rinf_c = om * rinf;
                    musum_c = fma(alpha, fma(alpha, e12[1], e12[0]), musum);
                }
            };
            bool corr = qlive;
            if constexpr (SKIPC) {
                const bool skip = qlive && sig3 < IPM_SKIP_SIGMA;
                if (__any(skip)) {   // (wave-uniform) the predictor step of these instances is their step
                    mx_dnu(rt, true);
                    if (skip) take_step(0, 0.0, fmin(1.0, frac * a_aff), c12);
                }
                corr = qlive && !skip;
                if (!__any(corr)) {   // nobody left for the corrector
                    PHW(8);
                    continue;
                }
            }
            PHW(4);
            // ---- corrector (same factorisation, vector sweep only)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const double v = vc(i) + dvc(dx, du, i);
                double dgi, ec;
                barrier_terms(i, v, 1, smu, dgi, ec);
                rt[i] = rg[i] + ec;
            }
            if constexpr (MX)
                mx_corr(rt, rb);
            else {
                PHW(5);
                if constexpr (SCAN) backward_scan(rt, rb); else backward<false>(Hs, rt, rb);
                PHW(6);
                if constexpr (SCAN) forward_scan(rb); else forward(rb);
                PHW(7);
            }
            rmax = 1.0;
            double d12[2] = {0.0, 0.0};   // sum lam t after the step = musum + alpha d12[0] + alpha^2 d12[1]
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (term && i < NU) continue;
                const double v = vc(i) + dvc(dx, du, i), dv = dvc(Dx, Du, i);
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    if (!has(sd, i)) continue;
                    double dt1, dl1, dt2, dl2, dss, rat;
                    row_steps(i, sd, v, dv, 1, smu, dt1, dl1, dt2, dl2, dss, rat);
                    rmax = fmax(rmax, rat);
                    d12[0] = fma(lam[sd][i], dt1, fma(t[sd][i], dl1, d12[0])), d12[1] = fma(dl1, dt1, d12[1]);
                    if (SOFT && softc(i)) {
                        const int ii = ss(i);
                        d12[0] = fma(lams[sd][ii], dt2, fma(ts[sd][ii], dl2, d12[0])), d12[1] = fma(dl2, dt2, d12[1]);
                    }
                }
            }
            seg_reduce<1, 2, M::SEG_SKIP>(&rmax, d12, k, lpi, base);
            if (corr) take_step(1, smu, fmin(1.0, frac * fast_rcp(rmax)), d12);
            PHW(8);
        }
        return ok;
    }

    // ---- sensitivities: dV/dp = dL/dp (nlp.py:1211,1401) and du0*/dp (nlp.py:1413-1424) by an adjoint solve --
    // nun: multiplier nu_{k+1} of the dynamics leaving this stage.  dVa / dpia: per-instance output rows.
    // hdd: sum_m nu_{k+1,m} hess F_m along the non-trivial coordinates, from linearize<true>() (only read when du0*/dp is wanted)
    MPCRL_DI void sensitivities(int flags, bool valid, const double *nun, double *dVa, double *dpia, const double *hdd) {
        // dF/dtheta of the stage: from a first-order evaluation of the map when only dV/dp is wanted; with du0*/dp it is the
        // first-order part of the mixed second-order evaluation below (one evaluation of the map less)
        const bool want_pi = (flags & 2) && dpia && !qmode;   // wave-uniform.  Q-mode: u_0 is pinned (mpc.py:71-76), du0/dp = 0
        double Fth[NX * NTD];
        if (!term && !want_pi) {
            Jet1<NTD> jx[NX], ju[NU], jt[NTD], jn[NX];
#pragma unroll
            for (int i = 0; i < NU; ++i) ju[i] = Jet1<NTD>(u[i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) jx[i] = Jet1<NTD>(x[i]);
#pragma unroll
            for (int i = 0; i < NTD; ++i) jt[i] = Jet1<NTD>(thd[i]), jt[i].d[i] = 1.0;
            disc_map<M, Jet1<NTD>>(jx, ju, jt, jn, sp.h, sp.rk_steps);
#pragma unroll
            for (int m = 0; m < NX; ++m)
#pragma unroll
                for (int d = 0; d < NTD; ++d) Fth[m * NTD + d] = jn[m].d[d];
        } else {
#pragma unroll
            for (int i = 0; i < NX * NTD; ++i) Fth[i] = 0.0;
        }
        auto emit_dV = [&]() {
            if (!((flags & 1) && dVa)) return;
            double cpart[NTC > 0 ? NTC : 1];
#pragma unroll
            for (int i = 0; i < (NTC > 0 ? NTC : 1); ++i) cpart[i] = 0.0;
            M::cost_dp(term, k, x, u, ck, cpart);
#pragma unroll
            for (int d = 0; d < NTD; ++d) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < NX; ++m) a = fma(nun[m], Fth[m * NTD + d], a);
                a = seg_sum<M::SEG_SKIP>(a, k, lpi, base);
                if (first && valid) dVa[M::td_index(d)] = a;
            }
#pragma unroll
            for (int d = 0; d < NTC; ++d) {
                const double a = seg_sum<M::SEG_SKIP>(cpart[d], k, lpi, base);
                if (first && valid) dVa[M::tc_index(d)] = a;
            }
        };
        if (!want_pi) {
            emit_dV();
            return;
        }
        // exact Lagrangian Hessian of the stage (nlp.py:1202,1224): c hess l + sum_m nu_{k+1,m} hess F_m
        double Hx[NW * (NW + 1) / 2];
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Hx[sym(i, j)] = ck * hess_of_stage(i, j);
        {
            // second derivatives of F vanish along the coordinates whose Jacobian columns are constant (M::lin_coord lists the others);
            // the rest came out of the linearisation's own evaluation of the map (linearize<true>; zero on the terminal lane)
            constexpr int ND = M::NLD;
#pragma unroll
            for (int di = 0; di < ND; ++di)
#pragma unroll
                for (int dj = 0; dj <= di; ++dj) Hx[sym(M::lin_coord(di), M::lin_coord(dj))] += hdd[di * (di + 1) / 2 + dj];
        }
        // barrier diagonal from the final (lam, t) of the BOUND rows; slacks are constants of the mirror (quirk q1)
        bool capped_x = false;      // the cap binds on a STATE row of this stage (an active state bound)
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            double d = 0.0;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
                if (has(sd, i)) {
                    const double w = lam[sd][i] / t[sd][i];
                    d += fmin(w, SENS_W_MAX);
                    if (i >= NU && w > SENS_W_MAX) capped_x = true;
                }
            Dg[i] = d;
        }
        auto Hs = [&](int i, int j) { return Hx[sym(i, j)]; };
        hscale = 1.0;
        double zero[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) zero[i] = 0.0;
        bool okall = true;
#pragma unroll
        for (int iu = 0; iu < NU; ++iu) {
#pragma unroll
            for (int i = 0; i < NW; ++i) rt[i] = (first && i == iu) ? -1.0 : 0.0;
            if constexpr (MX) {   // NU = 1: one adjoint solve
                const bool okf = mx_pred<true>(Hs, rt, zero);
                mx_dnu(rt, true);
                okall = seg_max<M::SEG_SKIP>(okf ? 0.0 : 1.0, k, lpi, base) < 0.5;
                // Richardson extrapolation in the stiffness cap (round 6).  Where the cap binds on a STATE row the capped solve is off by
                // c / W — 1.5e-6 ... 3.6e-6 of du0/dp on cartpole states with the cart at the end of its track, measured against
                // third-party finite differences (G7b) — linear in 1 / W down to W ~ 1e11, while the recursion's rounding grows with W
                // (table at SENS_W_MAX).  y = 2 y(W) - y(W / 2) takes the bias out at W's rounding level (3e-9 on the same states).
                // Decided per INSTANCE (an instance's numbers never depend on its wavefront neighbours); the second solve runs for the
                // wavefront when any of its instances needs it — none of the benchmark's, whose state bounds are inactive.
                const bool extrap = seg_max<M::SEG_SKIP>(capped_x ? 1.0 : 0.0, k, lpi, base) > 0.5;
                if (__any(extrap)) {
                    double y1x[NX], y1n[NX];
                    const double y1u = Du[0];
#pragma unroll
                    for (int i = 0; i < NX; ++i) y1x[i] = Dx[i], y1n[i] = Dnu[i];
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        double d = 0.0;
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd)
                            if (has(sd, i)) d += fmin(lam[sd][i] / t[sd][i], 0.5 * SENS_W_MAX);
                        Dg[i] = d;
                    }
                    const bool okf2 = mx_pred<true>(Hs, rt, zero);
                    mx_dnu(rt, true);
                    const bool ok2 = seg_max<M::SEG_SKIP>(okf2 ? 0.0 : 1.0, k, lpi, base) < 0.5;
                    okall = okall && (ok2 || !extrap);
                    Du[0] = extrap ? fma(2.0, y1u, -Du[0]) : y1u;
#pragma unroll
                    for (int i = 0; i < NX; ++i) Dx[i] = extrap ? fma(2.0, y1x[i], -Dx[i]) : y1x[i], Dnu[i] = extrap ? fma(2.0, y1n[i], -Dnu[i]) : y1n[i];
                }
            } else {
                if (iu == 0) {
                    const bool okf = backward<true>(Hs, rt, zero);
                    okall = seg_max<M::SEG_SKIP>(okf ? 0.0 : 1.0, k, lpi, base) < 0.5;
                } else if constexpr (SCAN)
                    backward_scan(rt, zero);
                else
                    backward<false>(Hs, rt, zero);
                if constexpr (SCAN) forward_scan(zero); else forward(zero);
            }
            double ynn[NX], yv[NW];
#pragma unroll
            for (int i = 0; i < NX; ++i) ynn[i] = lane_dn(Dnu[i]);
#pragma unroll
            for (int i = 0; i < NW; ++i) yv[i] = dvc(Dx, Du, i);
            double outd[NTD], outc[NTC > 0 ? NTC : 1];
#pragma unroll
            for (int d = 0; d < NTD; ++d) outd[d] = 0.0;
#pragma unroll
            for (int d = 0; d < (NTC > 0 ? NTC : 1); ++d) outc[d] = 0.0;
            M::cost_mixed(term, yv, ck, outc);
            if (!term) {
                Jet2<NTD> jx[NX], ju[NU], jt[NTD], jn[NX];
#pragma unroll
                for (int i = 0; i < NU; ++i) ju[i] = Jet2<NTD>(u[i]), ju[i].e = yv[i];
#pragma unroll
                for (int i = 0; i < NX; ++i) jx[i] = Jet2<NTD>(x[i]), jx[i].e = yv[NU + i];
#pragma unroll
                for (int i = 0; i < NTD; ++i) jt[i] = Jet2<NTD>(thd[i]), jt[i].g[i] = 1.0;
                disc_map<M, Jet2<NTD>>(jx, ju, jt, jn, sp.h, sp.rk_steps);
                if (iu == 0) {
#pragma unroll
                    for (int m = 0; m < NX; ++m)
#pragma unroll
                        for (int d = 0; d < NTD; ++d) Fth[m * NTD + d] = jn[m].g[d];
                }
#pragma unroll
                for (int d = 0; d < NTD; ++d) {
                    double a = 0.0;
#pragma unroll
                    for (int m = 0; m < NX; ++m) a = fma(nun[m], jn[m].m[d], fma(ynn[m], Fth[m * NTD + d], a));
                    outd[d] = a;
                }
            }
            if (iu == 0) emit_dV();
#pragma unroll
            for (int d = 0; d < NTD; ++d) {
                const double a = seg_sum<M::SEG_SKIP>(outd[d], k, lpi, base);
                if (first && valid) dpia[iu * NP + M::td_index(d)] = okall ? -a : NAN;
            }
#pragma unroll
            for (int d = 0; d < NTC; ++d) {
                const double a = seg_sum<M::SEG_SKIP>(outc[d], k, lpi, base);
                if (first && valid) dpia[iu * NP + M::tc_index(d)] = okall ? -a : NAN;
            }
        }
    }
};

// =====================================================================================================
// Sensitivity pass on the state a lane holds (x, u, nu, lam, t of its stage; parameters, bounds and cost table set): dV/dp and
// du0*/dp of the instance the lane's slot works on — what update_nlp computes after the reference's solve (nlp.py:1399-1424).
// Called by the solve kernels at the end of a wavefront's life (MPCRL_FUSE_SENS: "the adjoint KKT solve fused into the same
// sweep") and by small_sens_kernel on a stored iterate.  The output rows are written in full: the lanes of an instance zero
// every entry sensitivities() does not store — the cost block of p (zero gradient of the mirror, nlp.py:1039-1055), everything
// of an instance that was not solved, du0*/dp in Q-mode.
// =====================================================================================================
#ifndef MPCRL_FUSE_SENS
#define MPCRL_FUSE_SENS 0   // measured (cartpole, 4096): fused 0.582 ms vs 0.564 ms with the pass as a second launch — the second-order jets of the
                          // pass raise the register pressure of the SQP loop (sliced kernel: 36 -> 76 spilled VGPRs, 173 -> 237 SGPRs); the plain
                          // launch (3072 instances) gains 0.6 %.  The fused form is kept buildable and tested (-DMPCRL_FUSE_SENS=1).
#endif
template <class M>
MPCRL_DI void small_sens_tail(SmallSolver<M> &S, const SmallArgs &a, long inst, bool valid, int status) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int k = S.k, lpi = S.lpi;
    double xn[NX], nun[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = lane_dn(S.x[i]), nun[i] = lane_dn(S.nu_[i]);
    double hdd[M::NLD * (M::NLD + 1) / 2];
    if ((a.flags & 2) && a.dpi && !S.qmode) {   // wave-uniform
        S.template linearize<true>(xn, nun, hdd);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); if (S.pht) S.phw[9] += n_ - S.pht; S.pht = n_; }      // (phase profile of the sensitivity kernel)
#endif
    } else {
#pragma unroll
        for (int e = 0; e < M::NLD * (M::NLD + 1) / 2; ++e) hdd[e] = 0.0;
        S.linearize(xn);
    }
#pragma unroll
    for (int i = 0; i < S.NPK; ++i) S.P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) S.p[i] = 0.0;
    const bool sv = valid && (status == 0 || status == 2);
    S.sensitivities(a.flags, sv, nun, a.dV ? a.dV + inst * NP : nullptr, a.dpi ? a.dpi + inst * NU * NP : nullptr, hdd);
    // the zeros go last so that no load of the pass waits behind these stores; the two sets of addresses are disjoint
    if (valid) {
        if (a.dV)
            for (int e = k; e < NP; e += lpi)
                if (!(sv && M::p_has_gradient(e))) a.dV[inst * NP + e] = 0.0;
        if (a.dpi)
            for (int e = k; e < NU * NP; e += lpi)
                if (!(sv && !S.qmode && M::p_has_gradient(e % NP))) a.dpi[inst * NU * NP + e] = 0.0;
    }
}

// =====================================================================================================
// kernel: floor(64/(N+1)) instances per 64-lane workgroup
// =====================================================================================================
#ifndef MPCRL_LINEAR_OCC
#define MPCRL_LINEAR_OCC 2   // linear system (one instance per wavefront, N = 40): two wavefronts per SIMD measured faster than one
#endif
#ifndef MPCRL_CARTPOLE_OCC
#define MPCRL_CARTPOLE_OCC 1     // (2 + MPCRL_CARTPOLE_MAX_IPW=2 + MPCRL_MX_SLOTS=43: the two-wavefronts-per-SIMD experiment of round 6, profiles/README.md)
#endif
template <class M>
__global__ void __launch_bounds__(64, M::DISCRETE ? MPCRL_LINEAR_OCC : MPCRL_CARTPOLE_OCC) small_solve_kernel(const SmallSpec sp, const SmallArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NP = M::NP, NTD = M::NTD, NTC = M::NTC;
    constexpr bool SOFT = M::HAS_SOFT;
    const int lane = threadIdx.x;
    const int N = sp.N, lpi = N + 1, ipw = min(64 / lpi, M::MAX_IPW);
    const int slot = lane / lpi, k = lane - slot * lpi, base = slot * lpi;
    long inst = (long)blockIdx.x * ipw + slot;
    const bool valid = slot < ipw && inst < a.B;
    if (!valid) inst = a.B - 1;   // dead lanes shadow the last instance and never store
    if (a.perm) inst = a.perm[inst];
    SmallSolver<M> S(sp, k, lpi, base);
    __shared__ __attribute__((aligned(16))) double mx_lds[SmallSolver<M>::MX ? SmallSolver<M>::MX_LDS : 2];
    S.ms = mx_lds;
    const bool term = S.term, first = S.first;
    S.qmode = a.u0fix != nullptr;
    S.init_bounds();
    if (sp.cost_kind == 0)
        S.ck = term ? 1.0 : sp.dT;                                                        // nlp.py:1044-1055
    else
        S.ck = first ? sp.dT : (term ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);   // nlp.py:1083-1091
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    __shared__ double c_lds[M::MAX_IPW * SmallSolver<M>::CTAB];
    S.fill_cost_table(c_lds, slot < ipw ? slot : 0, slot < ipw, th);
    SmallSolver<M>::wave_lds_sync();
    S.load_hc();
    // (lanes past the last instance slot shadow another instance: they get a slot of their own)
    __shared__ double th_slots[SmallSolver<M>::TH_LDS ? (M::MAX_IPW + 1) * SmallSolver<M>::TH_DOUBLES : 1];
    S.bind_theta(th_slots + (slot < ipw ? slot : M::MAX_IPW) * SmallSolver<M>::TH_DOUBLES);
#pragma unroll
    for (int i = 0; i < NTD; ++i) S.thd.set(i, th[M::td_index(i)]);
#pragma unroll
    for (int i = 0; i < NTC; ++i) S.thc.set(i, th[M::tc_index(i)]);
    const double *x0 = a.x0 + inst * NX;
    const double *u0f = S.qmode ? a.u0fix + inst * NU : a.x0 + inst * NX;
#pragma unroll
    for (int i = 0; i < NX; ++i) S.x0r[i] = x0[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) S.u0r[i] = u0f[i];   // (not Q-mode: u0f aliases x0, never read)
    // ---- iterate: stored (warm) or the reference's cold start (MPC.reset, mpc.py:204-210)
    const size_t nb = (size_t)(N + 1) * NW;
    double *bnd = a.BND + (size_t)inst * 10 * nb + (size_t)k * NW;
    // the whole-batch flag is wave-uniform (a scalar branch); the per-instance mask is applied with selects inside the warm path —
    // a lane-divergent if/else around these stores makes the optimiser merge them through pointer phis and park state in scratch
    const bool cold = (a.flags & 8) || (a.cold && a.cold[inst]);
    if (a.flags & 8) {
#pragma unroll
        for (int i = 0; i < NX; ++i) S.x[i] = S.x0r[i], S.nu_[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) S.u[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            S.lam[0][i] = S.lam[1][i] = 0.0, S.t[0][i] = S.t[1][i] = 1.0, S.aff[0][i] = S.aff[1][i] = 0.0;
            if constexpr (SOFT) {
                if (M::soft_coord(i)) {
                    const int j = S.ss(i);
                    S.s[0][j] = S.s[1][j] = 0.0, S.lams[0][j] = S.lams[1][j] = 0.0, S.ts[0][j] = S.ts[1][j] = 1.0, S.affs[0][j] = S.affs[1][j] = 0.0;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double xs = a.X[(inst * (N + 1) + k) * NX + i], ns = a.PI[(inst * N + (first ? 0 : k - 1)) * NX + i];
            S.x[i] = cold ? S.x0r[i] : xs;
            S.nu_[i] = (first || cold) ? 0.0 : ns;
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const double us = a.U[(inst * N + (term ? 0 : k)) * NU + i];
            S.u[i] = (term || cold) ? 0.0 : us;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const double l0 = bnd[0 * nb + i], l1 = bnd[1 * nb + i], t0 = bnd[2 * nb + i], t1 = bnd[3 * nb + i];
            S.lam[0][i] = cold ? 0.0 : l0, S.lam[1][i] = cold ? 0.0 : l1, S.t[0][i] = cold ? 1.0 : t0, S.t[1][i] = cold ? 1.0 : t1;
            S.aff[0][i] = S.aff[1][i] = 0.0;
            if constexpr (SOFT) {
                if (M::soft_coord(i)) {
                    const int j = S.ss(i);
                    const double s0 = bnd[4 * nb + i], s1 = bnd[5 * nb + i], m0 = bnd[6 * nb + i], m1 = bnd[7 * nb + i], u0_ = bnd[8 * nb + i],
                                 u1_ = bnd[9 * nb + i];
                    S.s[0][j] = cold ? 0.0 : s0, S.s[1][j] = cold ? 0.0 : s1, S.lams[0][j] = cold ? 0.0 : m0, S.lams[1][j] = cold ? 0.0 : m1;
                    S.ts[0][j] = cold ? 1.0 : u0_, S.ts[1][j] = cold ? 1.0 : u1_, S.affs[0][j] = S.affs[1][j] = 0.0;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < S.NPK; ++i) S.P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) S.p[i] = 0.0, S.dx[i] = 0.0, S.nuq[i] = 0.0, S.Dx[i] = 0.0, S.Dnu[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) S.du[i] = 0.0, S.Du[i] = 0.0, S.kff[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) S.K[i] = 0.0;
#pragma unroll
    for (int i = 0; i < S.NLK; ++i) S.Li[i] = 0.0;

    // ---- full-step SQP (the reference requests no globalisation; config/cartpole.yaml:8-14)
    const bool rti = (a.flags & 4) != 0;
    const int max_iter = rti ? 1 : sp.max_iter;
    bool live = valid, last_tight = true;
    int status = 2, n_sqp = 0, n_ipm = 0;
    // opt-in divergence exit (mpcrl_set_exit_rule): best residual so far, its value at the last check, iterations to the next check
    double rbest = 1e300, rchk = 1e300;
    int exit_cnt = sp.exit_window;
    // size of the perturbation the next QP sees (< 0: nothing to start from): change of the pinned x0 / u0 for a warm call
    double stepn = -1.0;
    if (!(a.flags & (8 | 16))) {   // MPCRL_COLD_DUAL: stored primal iterate, interior point from its default point
        // (instances of the cold mask take part in the reduction and discard its result below)
        double sl = 0.0;
        if (first) {
#pragma unroll
            for (int i = 0; i < NX; ++i) sl = fmax(sl, fabs(S.x0r[i] - S.x[i]));
            if (S.qmode) {
#pragma unroll
                for (int i = 0; i < NU; ++i) sl = fmax(sl, fabs(S.u0r[i] - S.u[i]));
            }
        }
        stepn = seg_max<M::SEG_SKIP>(sl, k, lpi, base);
        if (cold) stepn = -1.0;
    }
    double Vout = 0.0, res_out[4] = {0, 0, 0, 0};
    double nun[NX];
    for (int it = 0;; ++it) {
        double xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = lane_dn(S.x[i]), nun[i] = lane_dn(S.nu_[i]);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); if (S.pht) S.phw[10] += n_ - S.pht; S.pht = n_; }
#endif
        const double cl = S.linearize(xn);
        double rl[4];
        S.nlp_res_local(nun, S.x0r, S.u0r, rl);
        double res[4] = {rl[0], rl[1], rl[2], rl[3]}, cost = cl;
        seg_reduce<4, 1, M::SEG_SKIP>(res, &cost, k, lpi, base);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); S.phw[9] += n_ - S.pht; S.pht = n_; }
#endif
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        if (live) {
            Vout = cost, n_sqp = it;
#pragma unroll
            for (int j = 0; j < 4; ++j) res_out[j] = res[j];
            if (!(rmax < 1e300) || !(fabs(cost) < 1e300))   // the max-reductions drop NaNs, the cost sum does not
                status = 1, live = false;
            else if (rmax < sp.tol && last_tight && !(rti && it == 0))
                status = 0, live = false;
            else if (it >= max_iter)
                status = rmax < sp.tol ? 0 : 2, live = false;
            else if (sp.exit_window > 0) {   // wave-uniform
                rbest = fmin(rbest, rmax);
                if (it == 0)
                    rchk = rmax;
                else if (--exit_cnt == 0) {
                    if (rbest > sp.exit_factor * rchk) status = 2, live = false;   // no progress over the window: give the lanes back
                    rchk = rbest, exit_cnt = sp.exit_window;
                }
            }
        }
        // QP tolerances of this iteration (per instance)
        const double rr_ = fmin(1.0, rmax), ad_ = (rmax < sp.tol || M::DISCRETE || M::EXACT_QP) ? 0.0 : IPM_ADAPT_C * rr_ * rr_;   // LQ model: first QP is the answer
        const double tol_res = fmin(IPM_ADAPT_CAP, fmax(IPM_TOL_RES, ad_)), tol_mu = fmin(0.1 * IPM_ADAPT_CAP, fmax(IPM_TOL_MU, 1e-2 * ad_));
        if (live) last_tight = tol_res <= IPM_TOL_RES && tol_mu <= IPM_TOL_MU;
        if (!__any(live)) break;
        const double warm_mu = (stepn < 0.0 || M::EXACT_QP) ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
        bool ok = S.qp_solve(live, S.x0r, S.u0r, n_ipm, warm_mu, tol_res, tol_mu);
        if constexpr (M::DISCRETE) {      // (LQ model: a failed WARM QP once more from the cold interior point — see lq_solve_kernel)
            const bool again = live && !ok && warm_mu > 0.0;
            if (__any(again)) {
                const bool o = S.qp_solve(again, S.x0r, S.u0r, n_ipm, 0.0, tol_res, tol_mu);
                if (again) ok = o;
            }
        }
        if (live && !ok) status = 4, live = false;
        {
            double sl = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) sl = fmax(sl, fabs(S.dx[i]));
            if (!term) {
#pragma unroll
                for (int i = 0; i < NU; ++i) sl = fmax(sl, fabs(S.du[i]));
            }
            stepn = seg_max<M::SEG_SKIP>(sl, k, lpi, base);
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < NX; ++i) S.x[i] += S.dx[i], S.nu_[i] = S.nuq[i];
#pragma unroll
            for (int i = 0; i < NU; ++i) S.u[i] += S.du[i];
        }
    }
#ifdef MPCRL_PROFILE_PHASES
    if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_ticks[i_], S.phw[i_]);
#endif
    // ---- results
    // Lagrangian of the mirror, L = cost + pi' g + lam' h (nlp.py:1180; MPC.get_L, mpc.py:325-332): g_k = F(x_k, u_k) - x_{k+1} is the
    // r of the last linearisation, h = -(slack of the bound row); slack rows -s <= 0 carry lams
    double lag = 0.0;
    if (a.LAG) {
        if (!term) {
#pragma unroll
            for (int m = 0; m < NX; ++m) lag = fma(nun[m], S.r[m], lag);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (term && i < NU) continue;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
                if (S.has(sd, i)) {
                    lag = fma(-S.lam[sd][i], S.bslack(sd, i, S.vc(i)), lag);
                    if constexpr (SOFT) {
                        if (S.softc(i)) lag = fma(-S.lams[sd][S.ss(i)], S.s[sd][S.ss(i)], lag);
                    }
                }
        }
        lag = seg_sum<M::SEG_SKIP>(lag, k, lpi, base);
    }
    if (valid && first) {
        if (a.LAG) a.LAG[inst] = Vout + lag;
#pragma unroll
        for (int i = 0; i < NU; ++i) a.u0_out[inst * NU + i] = S.u[i];
        a.V[inst] = Vout;
        a.status[inst] = status;
        if (a.iters) a.iters[inst * 2] = n_sqp, a.iters[inst * 2 + 1] = n_ipm;
#pragma unroll
        for (int j = 0; j < 4; ++j) a.RES[inst * 4 + j] = res_out[j];
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            a.X[(inst * (N + 1) + k) * NX + i] = S.x[i];
            if (!first) a.PI[(inst * N + k - 1) * NX + i] = S.nu_[i];
        }
        if (!term) {
#pragma unroll
            for (int i = 0; i < NU; ++i) a.U[(inst * N + k) * NU + i] = S.u[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            bnd[0 * nb + i] = S.has(0, i) ? S.lam[0][i] : 0.0, bnd[1 * nb + i] = S.has(1, i) ? S.lam[1][i] : 0.0;
            bnd[2 * nb + i] = S.has(0, i) ? S.t[0][i] : 1.0, bnd[3 * nb + i] = S.has(1, i) ? S.t[1][i] : 1.0;
            if constexpr (SOFT) {   // (coordinates the model cannot soften keep the values of a cold iterate)
                const bool sc = M::soft_coord(i);
                const int j = S.ss(i);
                bnd[4 * nb + i] = sc ? S.s[0][j] : 0.0, bnd[5 * nb + i] = sc ? S.s[1][j] : 0.0;
                bnd[6 * nb + i] = sc ? S.lams[0][j] : 0.0, bnd[7 * nb + i] = sc ? S.lams[1][j] : 0.0;
                bnd[8 * nb + i] = sc ? S.ts[0][j] : 1.0, bnd[9 * nb + i] = sc ? S.ts[1][j] : 1.0;
            }
        }
    }
#if MPCRL_FUSE_SENS
    // ---- sensitivities of the instances this wavefront solved, from the state its lanes still hold (wave-uniform branch)
    if (a.flags & (1 | 2)) small_sens_tail<M>(S, a, inst, valid, status);
#endif
}

// =====================================================================================================
// time-sliced variant of the solve kernel for COLD calls: a wavefront OWNS ipw + 1 instances and has lanes for ipw of them.
//
// One wavefront per SIMD is all the registers allow, and a wavefront has lanes for floor(64 / (N + 1)) instances (cartpole N = 20:
// 3), so a batch of 4096 is 1366 wavefronts on 1024 SIMDs: two wave lifetimes, the second one on a third of the chip.  Here the
// batch is cut into ceil(B / (ipw + 1)) wavefronts (4096 -> 1024: one round) and each wavefront advances its ipw + 1 instances one
// SQP iteration at a time, ipw of them per round: after every round one slot parks its instance (x, u, nu, lam, t go to the
// instance's stored-iterate arrays in HBM, which they are written to at the end of a solve anyway; a dozen scalars go to LDS) and
// takes the parked one.  All instances progress at ipw / (ipw + 1) of full speed, so the wavefront lives (ipw + 1) / ipw as long
// and the launch takes 1.33 instead of 2 wave lifetimes.  The exchange stays inside one wavefront — its memory operations are
// performed in order, a wavefront-scope fence is all the ordering needed (as in chain_kernel.hpp) — which is what the cross-wavefront
// attempt of round 1 (claims by compare-and-swap) could not have.  An instance that finishes is written out at once and
// its slot goes to the parked instance for good.  The iteration of an instance is bit-for-bit the one of small_solve_kernel.
// =====================================================================================================
// LDSPARK: where the parked instance lives (chosen by the host from the horizon: sliced_parks_in_lds); a template parameter so that
// the kernel of the short horizons carries neither the code nor the pointers of the HBM exchange.
template <class M>
constexpr int sliced_park_words() { return 2 * M::NX + M::NU + 4 * (M::NX + M::NU); }
template <class M>
constexpr int sliced_park_cap() { return (M::NX == 4 && M::NU == 1) ? 21 * sliced_park_words<M>() : 64 * sliced_park_words<M>(); }
template <class M>
inline bool sliced_parks_in_lds(int N) { return (N + 1) * sliced_park_words<M>() <= sliced_park_cap<M>(); }

// WARM: the instances start from their stored iterates (and the per-instance cold mask), as in small_solve_kernel — a separate
// instantiation, so that the cold one keeps its register allocation.
template <class M, bool LDSPARK, bool WARM = false>
__global__ void __launch_bounds__(64, 1) small_solve_sliced_kernel(const SmallSpec sp, const SmallArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NP = M::NP, NTD = M::NTD, NTC = M::NTC;
    static_assert(!M::HAS_SOFT, "hard bounds only");
    const int lane = threadIdx.x;
    const int N = sp.N, lpi = N + 1, ipw = min(64 / lpi, M::MAX_IPW - 1), q = ipw + 1;
    const int slot = lane / lpi, k = lane - slot * lpi, base = slot * lpi;
    // Instance l of wavefront w is entry l * (number of wavefronts) + w of the packing order: a wavefront takes one instance from each
    // q-quantile of the order.  The order puts similar problems side by side, and with consecutive entries the wavefronts of the hard
    // end of the batch need more rounds than the others — with a single round of wavefronts the launch lasts as long as the slowest.
    const long nwave = gridDim.x;
    auto posof = [&](int l) { return (long)l * nwave + blockIdx.x; };
    const bool slot_on = slot < ipw;
    // ---- binding of this lane to an instance: changes when its slot takes the parked instance
    int loc = slot_on ? slot : ipw;                   // local index; the first lane past the slots speaks for the parked instance
    long inst = 0;
    bool valid = false;
    const double *x0 = nullptr, *u0f = nullptr, *th = nullptr;
    double *bnd = nullptr;
    const size_t nb = (size_t)(N + 1) * NW;
    SmallSolver<M> S(sp, k, lpi, base);
    __shared__ __attribute__((aligned(16))) double mx_lds[SmallSolver<M>::MX ? SmallSolver<M>::MX_LDS : 2];
    __shared__ double c_lds[M::MAX_IPW * SmallSolver<M>::CTAB];
    __shared__ double sc_lds[M::MAX_IPW * 6];         // parked scalars: live, status, iterations, interior-point iterations, ...
    // The parked instance's state (x, nu, u, lam, t of every stage), field-major so that the lanes of a slot touch consecutive
    // words.  It fits next to the stage slots of the matrix-core sweep up to N + 1 = 21 stages (cartpole's horizon: 609 doubles —
    // the LDS of a CU is shared by four wavefronts, 40 960 B each); longer horizons park in the instance's stored-iterate arrays
    // in HBM, which costs a global round trip per round (3.3 us of a 45 us round: 8 % of the launch).
    constexpr int PK = 2 * NX + NU + 4 * NW;
    static_assert(PK == sliced_park_words<M>(), "host and kernel agree on the parked state");
    constexpr int PARK_CAP = LDSPARK ? sliced_park_cap<M>() : 1;
    __shared__ double park_lds[PARK_CAP];
    constexpr bool lds_park = LDSPARK;
    S.ms = mx_lds;
    S.qmode = a.u0fix != nullptr;
    S.init_bounds();
    // Batch index (after the packing order) and dynamics parameters of the ipw + 1 instances, looked up ONCE: a rebinding between
    // rounds must not cost global round trips of its own (perm -> theta -> state would be three in a row).
    __shared__ long gi_lds[M::MAX_IPW];
    constexpr int TH = NTD + NTC > 0 ? NTD + NTC : 1;
    __shared__ double th_lds[M::MAX_IPW * TH];
    {
        long i = posof(loc);
        if (i >= a.B) i = a.B - 1;                    // lanes without an instance shadow the last one and never store
        if (a.perm) i = a.perm[i];
        inst = i;
        th = a.theta + (size_t)i * a.theta_stride;
        if (k == 0 && slot <= ipw) {                  // the stage-0 lane of every slot, and the first lane past the slots (parked instance)
            gi_lds[loc] = i;
#pragma unroll
            for (int j = 0; j < NTD; ++j) th_lds[loc * TH + j] = th[M::td_index(j)];
#pragma unroll
            for (int j = 0; j < NTC; ++j) th_lds[loc * TH + NTD + j] = th[M::tc_index(j)];
        }
    }
    const bool term = S.term, first = S.first;
    if (sp.cost_kind == 0)
        S.ck = term ? 1.0 : sp.dT;
    else
        S.ck = first ? sp.dT : (term ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);
    // cost tables of all ipw + 1 instances (the parked one's by the first lane past the slots, which is a stage-0 lane)
    S.fill_cost_table(c_lds, loc, (slot_on || (slot == ipw && k == 0)) && posof(loc) < a.B, th);
    SmallSolver<M>::wave_lds_sync();
    auto bind = [&](int l) {
        valid = slot_on && posof(l) < a.B;
        const long i = gi_lds[l];
        inst = i;
        x0 = a.x0 + i * NX;
        u0f = S.qmode ? a.u0fix + i * NU : a.x0 + i * NX;
        bnd = a.BND + (size_t)i * 10 * nb + (size_t)k * NW;
    };
    if (!slot_on) loc = 0;                            // from here on the lanes past the slots shadow slot 0's instance
    bind(loc);
    auto load_params = [&]() {
        S.ctab = c_lds + loc * SmallSolver<M>::CTAB + S.stage_kind() * SmallSolver<M>::CSET;
        S.load_hc();
#pragma unroll
        for (int i = 0; i < NTD; ++i) S.thd.set(i, th_lds[loc * TH + i]);
#pragma unroll
        for (int i = 0; i < NTC; ++i) S.thc.set(i, th_lds[loc * TH + NTD + i]);
    };
    load_params();
    // ---- cold start of the resident instances (MPC.reset, mpc.py:204-210), or their stored iterates
#pragma unroll
    for (int i = 0; i < NX; ++i) S.x0r[i] = x0[i], S.x[i] = S.x0r[i], S.nu_[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) S.u[i] = 0.0, S.u0r[i] = u0f[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) S.lam[0][i] = S.lam[1][i] = 0.0, S.t[0][i] = S.t[1][i] = 1.0, S.aff[0][i] = S.aff[1][i] = 0.0;
    // stored state of the lane's current instance, with the per-instance cold mask applied by selects (small_solve_kernel's warm path)
    auto load_stored = [&]() {
        const bool cold = (a.flags & 8) || (a.cold && a.cold[inst]);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double xs = a.X[(inst * (N + 1) + k) * NX + i], ns = a.PI[(inst * N + (first ? 0 : k - 1)) * NX + i];
            S.x[i] = cold ? S.x0r[i] : xs;
            S.nu_[i] = (first || cold) ? 0.0 : ns;
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const double us = a.U[(inst * N + (term ? 0 : k)) * NU + i];
            S.u[i] = (term || cold) ? 0.0 : us;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const double l0 = bnd[0 * nb + i], l1 = bnd[1 * nb + i], t0 = bnd[2 * nb + i], t1 = bnd[3 * nb + i];
            S.lam[0][i] = cold ? 0.0 : l0, S.lam[1][i] = cold ? 0.0 : l1, S.t[0][i] = cold ? 1.0 : t0, S.t[1][i] = cold ? 1.0 : t1;
        }
    };
    // change of the pinned x0 / u0 against the stored iterate: the perturbation a warm call's first QP sees (< 0: start cold)
    auto warm_stepn = [&]() {
        double sl = 0.0;
        if (first) {
#pragma unroll
            for (int i = 0; i < NX; ++i) sl = fmax(sl, fabs(S.x0r[i] - S.x[i]));
            if (S.qmode) {
#pragma unroll
                for (int i = 0; i < NU; ++i) sl = fmax(sl, fabs(S.u0r[i] - S.u[i]));
            }
        }
        const double sn = seg_max<M::SEG_SKIP>(sl, k, lpi, base);
        return ((a.flags & (8 | 16)) || (a.cold && a.cold[inst])) ? -1.0 : sn;
    };
    if constexpr (WARM) load_stored();
#pragma unroll
    for (int i = 0; i < S.NPK; ++i) S.P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) S.p[i] = 0.0, S.dx[i] = 0.0, S.nuq[i] = 0.0, S.Dx[i] = 0.0, S.Dnu[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) S.du[i] = 0.0, S.Du[i] = 0.0, S.kff[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) S.K[i] = 0.0;
#pragma unroll
    for (int i = 0; i < S.NLK; ++i) S.Li[i] = 0.0;
    const int max_iter = sp.max_iter;
    // per-instance scalars of the resident instance (uniform over the lanes of a slot)
    bool live = valid, last_tight = true;
    int status = 2, iti = 0, n_ipm = 0;
    double stepn = -1.0;
    if constexpr (WARM) stepn = warm_stepn();
    double rbest = 1e300, rchk = 1e300;               // divergence exit (mpcrl_set_exit_rule), as in small_solve_kernel
    int exit_cnt = sp.exit_window;
    // the parked instance (wave-uniform): local index or -1, and whether it has run at all
    int pk = (posof(ipw) < a.B) ? ipw : -1;
    bool pk_started = false;
    int rr = 0, rounds_since = 0;
    int first_done = -1;                              // local index of the instance that left its slot for good (wave-uniform), or -1
    double nun[NX];
    for (;;) {
        double xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = lane_dn(S.x[i]), nun[i] = lane_dn(S.nu_[i]);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); if (S.pht) S.phw[15] += n_ - S.pht; S.pht = n_; }
#endif
        const double cl = S.linearize(xn);
        double rl[4];
        S.nlp_res_local(nun, S.x0r, S.u0r, rl);
        double res[4] = {rl[0], rl[1], rl[2], rl[3]}, cost = cl;
        seg_reduce<4, 1, M::SEG_SKIP>(res, &cost, k, lpi, base);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); S.phw[9] += n_ - S.pht; S.pht = n_; }
#endif
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        bool fin_now = false;
        if (live) {
            if (!(rmax < 1e300) || !(fabs(cost) < 1e300))   // the max-reductions drop NaNs, the cost sum does not
                status = 1, live = false;
            else if (rmax < sp.tol && last_tight)
                status = 0, live = false;
            else if (iti >= max_iter)
                status = rmax < sp.tol ? 0 : 2, live = false;
            else if (sp.exit_window > 0) {   // wave-uniform
                rbest = fmin(rbest, rmax);
                if (iti == 0)
                    rchk = rmax;
                else if (--exit_cnt == 0) {
                    if (rbest > sp.exit_factor * rchk) status = 2, live = false;
                    rchk = rbest, exit_cnt = sp.exit_window;
                }
            }
            fin_now = !live;
        }
        if (__any(fin_now)) {   // ---- results + iterate of the instances that finish here (wave-uniform branch: the reduction inside is safe)
            double lag = 0.0;
            if (a.LAG) {
                if (!term) {
#pragma unroll
                    for (int m = 0; m < NX; ++m) lag = fma(nun[m], S.r[m], lag);
                }
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (term && i < NU) continue;
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (S.has(sd, i)) lag = fma(-S.lam[sd][i], S.bslack(sd, i, S.vc(i)), lag);
                }
                lag = seg_sum<M::SEG_SKIP>(lag, k, lpi, base);
            }
            if (fin_now && valid) {
                if (first) {
                    if (a.LAG) a.LAG[inst] = cost + lag;
#pragma unroll
                    for (int i = 0; i < NU; ++i) a.u0_out[inst * NU + i] = S.u[i];
                    a.V[inst] = cost;
                    a.status[inst] = status;
                    if (a.iters) a.iters[inst * 2] = iti, a.iters[inst * 2 + 1] = n_ipm;
#pragma unroll
                    for (int j = 0; j < 4; ++j) a.RES[inst * 4 + j] = res[j];
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    a.X[(inst * (N + 1) + k) * NX + i] = S.x[i];
                    if (!first) a.PI[(inst * N + k - 1) * NX + i] = S.nu_[i];
                }
                if (!term) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) a.U[(inst * N + k) * NU + i] = S.u[i];
                }
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    bnd[0 * nb + i] = S.has(0, i) ? S.lam[0][i] : 0.0, bnd[1 * nb + i] = S.has(1, i) ? S.lam[1][i] : 0.0;
                    bnd[2 * nb + i] = S.has(0, i) ? S.t[0][i] : 1.0, bnd[3 * nb + i] = S.has(1, i) ? S.t[1][i] : 1.0;
                }
            }
        }
        const double rr_ = fmin(1.0, rmax), ad_ = rmax < sp.tol ? 0.0 : IPM_ADAPT_C * rr_ * rr_;
        const double tol_res = fmin(IPM_ADAPT_CAP, fmax(IPM_TOL_RES, ad_)), tol_mu = fmin(0.1 * IPM_ADAPT_CAP, fmax(IPM_TOL_MU, 1e-2 * ad_));
        if (live) last_tight = tol_res <= IPM_TOL_RES && tol_mu <= IPM_TOL_MU;
        const bool any_live = __any(live);
        if (!any_live && pk < 0) break;
        if (any_live) {
            const double warm_mu = stepn < 0.0 ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
            const bool ok = S.qp_solve(live, S.x0r, S.u0r, n_ipm, warm_mu, tol_res, tol_mu);
            bool qp_failed = live && !ok;
            {
                double sl = 0.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) sl = fmax(sl, fabs(S.dx[i]));
                if (!term) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) sl = fmax(sl, fabs(S.du[i]));
                }
                const double sn = seg_max<M::SEG_SKIP>(sl, k, lpi, base);
                if (live) stepn = sn;
            }
            if (live && ok) {
#pragma unroll
                for (int i = 0; i < NX; ++i) S.x[i] += S.dx[i], S.nu_[i] = S.nuq[i];
#pragma unroll
                for (int i = 0; i < NU; ++i) S.u[i] += S.du[i];
                ++iti;
            }
            if (__any(qp_failed)) {   // QP failure: status 4 with the last iterate (the same outputs small_solve_kernel leaves)
                double cst = cost;
                if (qp_failed && valid) {
                    if (first) {
                        if (a.LAG) a.LAG[inst] = cst;
#pragma unroll
                        for (int i = 0; i < NU; ++i) a.u0_out[inst * NU + i] = S.u[i];
                        a.V[inst] = cst;
                        a.status[inst] = 4;
                        if (a.iters) a.iters[inst * 2] = iti, a.iters[inst * 2 + 1] = n_ipm;
#pragma unroll
                        for (int j = 0; j < 4; ++j) a.RES[inst * 4 + j] = res[j];
                    }
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        a.X[(inst * (N + 1) + k) * NX + i] = S.x[i];
                        if (!first) a.PI[(inst * N + k - 1) * NX + i] = S.nu_[i];
                    }
                    if (!term) {
#pragma unroll
                        for (int i = 0; i < NU; ++i) a.U[(inst * N + k) * NU + i] = S.u[i];
                    }
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        bnd[0 * nb + i] = S.has(0, i) ? S.lam[0][i] : 0.0, bnd[1 * nb + i] = S.has(1, i) ? S.lam[1][i] : 0.0;
                        bnd[2 * nb + i] = S.has(0, i) ? S.t[0][i] : 1.0, bnd[3 * nb + i] = S.has(1, i) ? S.t[1][i] : 1.0;
                    }
                }
                if (qp_failed) status = 4, live = false;
            }
        }
        if (pk < 0) continue;
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); S.phw[10] += n_ - S.pht; S.pht = n_; }
#endif
        // ---- rotation: one slot hands its instance over to the parked one.  The slot of a finished instance first (for good),
        // else round robin.  Every decision below is a function of wave-uniform values.
        int v = -1;
        for (int s_ = 0; s_ < ipw; ++s_)
            if (v < 0 && !__shfl(live ? 1 : 0, s_ * lpi)) v = s_;
        const bool for_good = v >= 0;
#ifndef MPCRL_SLICE_PERIOD
#define MPCRL_SLICE_PERIOD 1
#endif
        if (!for_good && (++rounds_since % MPCRL_SLICE_PERIOD) != 0) continue;   // rotate every MPCRL_SLICE_PERIOD-th round only
        if (v < 0) v = rr % ipw, ++rr;
        const bool sw = slot_on && slot == v;
        const int lo = __shfl(loc, v * lpi);            // local index of the outgoing instance
#if MPCRL_FUSE_SENS
        // an instance that leaves its slot for good (it finished) is remembered: with the park area in LDS its final state goes
        // there in place of the parked one's, so that the wavefront can run its sensitivity pass at the end without reading it back
        const bool keep_out = !for_good || (lds_park && (a.flags & 3));
        if (for_good) first_done = lo;
#else
        const bool keep_out = !for_good;
#endif
        if (sw && !for_good) {                          // park: state to LDS (or the instance's stored-iterate arrays), scalars to LDS
            if (valid && !lds_park) {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    a.X[(inst * (N + 1) + k) * NX + i] = S.x[i];
                    if (!first) a.PI[(inst * N + k - 1) * NX + i] = S.nu_[i];
                }
                if (!term) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) a.U[(inst * N + k) * NU + i] = S.u[i];
                }
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    bnd[0 * nb + i] = S.lam[0][i], bnd[1 * nb + i] = S.lam[1][i], bnd[2 * nb + i] = S.t[0][i], bnd[3 * nb + i] = S.t[1][i];
            }
            if (first) {   // the small integers share one word (exact in a double): live | tight | status (3 bits) | countdown (8) | iterations
                double *sc = sc_lds + lo * 6;
                sc[0] = (double)((live ? 1 : 0) + (last_tight ? 2 : 0) + 4 * status + 32 * (exit_cnt & 255)) + 8192.0 * (double)iti;
                sc[1] = (double)n_ipm, sc[2] = stepn, sc[3] = rbest, sc[4] = rchk;
            }
        }
#if MPCRL_FUSE_SENS
        if (sw && for_good && first) sc_lds[lo * 6 + 5] = (double)status;   // (read again by the sensitivity pass at the end)
#endif
        SmallSolver<M>::wave_lds_sync();                // orders the stores above before the loads below (same wavefront: in order)
        // take the parked instance: every lane runs the same loads, the lanes of the slot keep the results
        const int newloc = sw ? pk : loc;
        loc = newloc;
        bind(loc);
        load_params();
        {
            const double *sc = sc_lds + pk * 6;
            const double w_ = sc[0], ni_ = sc[1], sn_ = sc[2], rb_ = sc[3], rc_ = sc[4];
            const int it_ = (int)(w_ * (1.0 / 8192.0)), lo_ = (int)(w_ - 8192.0 * (double)it_);   // iterations | the packed low bits
            if (sw) {
                live = pk_started ? ((lo_ & 1) != 0) : valid;
                last_tight = pk_started ? ((lo_ & 2) != 0) : true;
                status = pk_started ? ((lo_ >> 2) & 7) : 2;
                exit_cnt = pk_started ? ((lo_ >> 5) & 255) : sp.exit_window;
                iti = pk_started ? it_ : 0;
                n_ipm = pk_started ? (int)ni_ : 0;
                stepn = pk_started ? sn_ : -1.0;
                rbest = pk_started ? rb_ : 1e300;
                rchk = pk_started ? rc_ : 1e300;
            }
        }
        if constexpr (lds_park) {
            // the lanes of the slot swap their registers with the parked state (lane k <-> column k: no lane reads what another
            // one writes); a parked instance that has not run yet starts cold
            double *pl = park_lds + k;
            double xc[NX];   // x0 of the incoming instance: a global read, on its first take only (wave-uniform branch)
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = 0.0;
            if (!pk_started) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = x0[i];
            }
            // all reads first, then all writes: interleaved, every read would wait for the write before it (same array)
            double in[PK], out[PK];
#pragma unroll
            for (int j = 0; j < PK; ++j) in[j] = pl[j * lpi];
#pragma unroll
            for (int i = 0; i < NX; ++i) out[i] = S.x[i], out[NX + i] = first ? S.x0r[i] : S.nu_[i];   // nu of stage 0 is identically zero: its words carry x0
#pragma unroll
            for (int i = 0; i < NU; ++i) out[2 * NX + i] = S.u[i];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                out[2 * NX + NU + 4 * i] = S.lam[0][i], out[2 * NX + NU + 4 * i + 1] = S.lam[1][i];
                out[2 * NX + NU + 4 * i + 2] = S.t[0][i], out[2 * NX + NU + 4 * i + 3] = S.t[1][i];
            }
            if (sw && keep_out) {
#pragma unroll
                for (int j = 0; j < PK; ++j) pl[j * lpi] = out[j];
            }
            if (sw) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    S.x[i] = pk_started ? in[i] : xc[i], S.nu_[i] = (pk_started && !first) ? in[NX + i] : 0.0, S.x0r[i] = pk_started ? in[NX + i] : xc[i];
#pragma unroll
                for (int i = 0; i < NU; ++i) S.u[i] = pk_started ? in[2 * NX + i] : 0.0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    S.lam[0][i] = pk_started ? in[2 * NX + NU + 4 * i] : 0.0, S.lam[1][i] = pk_started ? in[2 * NX + NU + 4 * i + 1] : 0.0;
                    S.t[0][i] = pk_started ? in[2 * NX + NU + 4 * i + 2] : 1.0, S.t[1][i] = pk_started ? in[2 * NX + NU + 4 * i + 3] : 1.0;
                }
            }
            if constexpr (WARM) {
                if (!pk_started) {                    // (wave-uniform) first take of the parked instance: its stored iterate, once
                    double kx[NX], kn[NX], ku[NU], kl[2][NW], kt[2][NW];
#pragma unroll
                    for (int i = 0; i < NX; ++i) kx[i] = S.x[i], kn[i] = S.nu_[i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) ku[i] = S.u[i];
#pragma unroll
                    for (int i = 0; i < NW; ++i) kl[0][i] = S.lam[0][i], kl[1][i] = S.lam[1][i], kt[0][i] = S.t[0][i], kt[1][i] = S.t[1][i];
                    load_stored();
                    if (!sw) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) S.x[i] = kx[i], S.nu_[i] = kn[i];
#pragma unroll
                        for (int i = 0; i < NU; ++i) S.u[i] = ku[i];
#pragma unroll
                        for (int i = 0; i < NW; ++i) S.lam[0][i] = kl[0][i], S.lam[1][i] = kl[1][i], S.t[0][i] = kt[0][i], S.t[1][i] = kt[1][i];
                    }
                }
            }
        } else {
            // the arrays hold the parked state — or, on a warm call's first take, the stored iterate the instance starts from
            const bool from_arrays = pk_started || (WARM && !(a.flags & 8) && !(a.cold && a.cold[inst]));
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double xs = a.X[(inst * (N + 1) + k) * NX + i], ns = a.PI[(inst * N + (first ? 0 : k - 1)) * NX + i];
                const double x0n = x0[i];
                S.x[i] = sw ? (from_arrays ? xs : x0n) : S.x[i];
                S.x0r[i] = sw ? x0n : S.x0r[i];
                S.nu_[i] = sw ? ((first || !from_arrays) ? 0.0 : ns) : S.nu_[i];
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const double us = a.U[(inst * N + (term ? 0 : k)) * NU + i];
                S.u[i] = sw ? ((term || !from_arrays) ? 0.0 : us) : S.u[i];
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const double l0 = bnd[0 * nb + i], l1 = bnd[1 * nb + i], t0 = bnd[2 * nb + i], t1 = bnd[3 * nb + i];
                S.lam[0][i] = sw ? (from_arrays ? l0 : 0.0) : S.lam[0][i], S.lam[1][i] = sw ? (from_arrays ? l1 : 0.0) : S.lam[1][i];
                S.t[0][i] = sw ? (from_arrays ? t0 : 1.0) : S.t[0][i], S.t[1][i] = sw ? (from_arrays ? t1 : 1.0) : S.t[1][i];
            }
        }
        if (S.qmode) {   // the pinned u0 of the incoming instance (Q-mode only: one global read per rotation)
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const double un = u0f[i];
                S.u0r[i] = sw ? un : S.u0r[i];
            }
        }
        if constexpr (WARM) {
            if (!pk_started) {   // (wave-uniform) the incoming instance's first QP: every lane takes part in the reduction, its slot keeps the result
                const double sn = warm_stepn();
                if (sw) stepn = sn;
            }
        }
        pk = for_good ? -1 : lo;
        pk_started = true;
    }
#ifdef MPCRL_PROFILE_PHASES
    if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_ticks[i_], S.phw[i_]);
#endif
#if MPCRL_FUSE_SENS
    // ---- sensitivities.  Pass 1: the instances the slots hold at the end, from registers.  Pass 2: the instance that left its slot
    // for good earlier, on slot 0 from the park area (LDS parking) — or, with HBM parking, from its stored iterate, which this
    // wavefront wrote itself (its own stores are visible to its later loads once they have been waited for).
    if (a.flags & 3) {
        small_sens_tail<M>(S, a, inst, valid, status);
        if (first_done >= 0) {
            const bool on0 = slot == 0;
            loc = first_done;
            bind(loc);
            valid = on0 && posof(loc) < a.B;
            load_params();
            const int st_ = (int)sc_lds[first_done * 6 + 5];
            if constexpr (lds_park) {
                const double *pl = park_lds + k;
                double in[PK];
#pragma unroll
                for (int j = 0; j < PK; ++j) in[j] = pl[j * lpi];
#pragma unroll
                for (int i = 0; i < NX; ++i) S.x[i] = in[i], S.nu_[i] = first ? 0.0 : in[NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) S.u[i] = in[2 * NX + i];
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    S.lam[0][i] = in[2 * NX + NU + 4 * i], S.lam[1][i] = in[2 * NX + NU + 4 * i + 1];
                    S.t[0][i] = in[2 * NX + NU + 4 * i + 2], S.t[1][i] = in[2 * NX + NU + 4 * i + 3];
                }
            } else {
                __builtin_amdgcn_s_waitcnt(0);   // the stores of the finished instance have completed
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    S.x[i] = a.X[(inst * (N + 1) + k) * NX + i];
                    S.nu_[i] = first ? 0.0 : a.PI[(inst * N + k - 1) * NX + i];
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) S.u[i] = term ? 0.0 : a.U[(inst * N + k) * NU + i];
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    S.lam[0][i] = bnd[0 * nb + i], S.lam[1][i] = bnd[1 * nb + i], S.t[0][i] = bnd[2 * nb + i], S.t[1][i] = bnd[3 * nb + i];
            }
            small_sens_tail<M>(S, a, inst, valid, st_);
        }
    }
#endif
}

// =====================================================================================================
// sensitivity kernel: re-reads the converged iterate (x, u, nu, lam, t) written by small_solve_kernel.  With MPCRL_FUSE_SENS the
// solve kernels run the pass themselves at the end of a wavefront's life and this kernel is not launched (it stays for builds
// with MPCRL_FUSE_SENS = 0, where the pass was a second launch: 49 us + a 17 us boundary per 4096 cartpole instances).
// =====================================================================================================
template <class M>
__global__ void __launch_bounds__(64) small_sens_kernel(const SmallSpec sp, const SmallArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NP = M::NP, NTD = M::NTD, NTC = M::NTC;
    const int lane = threadIdx.x;
    const int N = sp.N, lpi = N + 1, ipw = min(64 / lpi, M::MAX_IPW);
    const int slot = lane / lpi, k = lane - slot * lpi, base = slot * lpi;
    long inst = (long)blockIdx.x * ipw + slot;
    const bool valid = slot < ipw && inst < a.B;
    if (!valid) inst = a.B - 1;
    if (a.perm) inst = a.perm[inst];
    SmallSolver<M> S(sp, k, lpi, base);
    __shared__ __attribute__((aligned(16))) double mx_lds[SmallSolver<M>::MX ? SmallSolver<M>::MX_LDS : 2];
    S.ms = mx_lds;
    const bool term = S.term, first = S.first;
    S.qmode = a.u0fix != nullptr;
    S.init_bounds();
    if (sp.cost_kind == 0)
        S.ck = term ? 1.0 : sp.dT;
    else
        S.ck = first ? sp.dT : (term ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    __shared__ double c_lds[M::MAX_IPW * SmallSolver<M>::CTAB];
    S.fill_cost_table(c_lds, slot < ipw ? slot : 0, slot < ipw, th);
    SmallSolver<M>::wave_lds_sync();
    S.load_hc();
    // (lanes past the last instance slot shadow another instance: they get a slot of their own)
    __shared__ double th_slots[SmallSolver<M>::TH_LDS ? (M::MAX_IPW + 1) * SmallSolver<M>::TH_DOUBLES : 1];
    S.bind_theta(th_slots + (slot < ipw ? slot : M::MAX_IPW) * SmallSolver<M>::TH_DOUBLES);
#pragma unroll
    for (int i = 0; i < NTD; ++i) S.thd.set(i, th[M::td_index(i)]);
#pragma unroll
    for (int i = 0; i < NTC; ++i) S.thc.set(i, th[M::tc_index(i)]);
    const size_t nb = (size_t)(N + 1) * NW;
    const double *bnd = a.BND + (size_t)inst * 10 * nb + (size_t)k * NW;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        S.x[i] = a.X[(inst * (N + 1) + k) * NX + i];
        S.nu_[i] = first ? 0.0 : a.PI[(inst * N + k - 1) * NX + i];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) S.u[i] = term ? 0.0 : a.U[(inst * N + k) * NU + i];
#pragma unroll
    for (int i = 0; i < NW; ++i) S.lam[0][i] = bnd[0 * nb + i], S.lam[1][i] = bnd[1 * nb + i], S.t[0][i] = bnd[2 * nb + i], S.t[1][i] = bnd[3 * nb + i];
    // the dynamics Jacobians of the final iterate are needed by the adjoint Riccati sweep: linearised again inside the pass
#ifdef MPCRL_PROFILE_PHASES
    S.pht = clock64();
#endif
    small_sens_tail<M>(S, a, inst, valid, a.status[inst]);
#ifdef MPCRL_PROFILE_PHASES
    { unsigned long long n_ = clock64(); S.phw[8] += n_ - S.pht; }
    if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_ticks[i_], S.phw[i_]);
#endif
}

}  // namespace mpcrl
