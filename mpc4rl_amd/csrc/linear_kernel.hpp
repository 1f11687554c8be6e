// linear_kernel.hpp — the linear-system OCP (nx = 2, nu = 1, N = 40; rlmpc/mpc/linear_system/acados.py:73-131) with SEVERAL STAGES PER LANE.
//
// small_solve_kernel gives every shooting stage a lane: 41 lanes, one instance per wavefront, and the serial Riccati factor sweep — 60 %
// of its instructions — runs with ONE useful lane of 64.  Its SQ counters (profiles/r05_small_pmc.txt): 103 k wave-instructions per
// instance, 66 % of the VALU issue floor of that stream: only instances SHARING wave-instructions can make it faster.  Here a lane
// holds SPL = 3 consecutive stages (stage = 3 pos + j), an instance takes ceil((N + 1) / 3) = 14 lanes and a wavefront FOUR instances,
// each in a DPP row (16 lanes) of its own (RL = 16; up to 8 lanes per instance: half rows, eight instances; horizons whose three-stage
// lanes would outgrow a row — N >= 48 — take FOUR stages per lane, SPL = 4, and fit a row again): 4096 instances are 1024 wavefronts,
// one per SIMD, one round.
//   * factor sweep: still N + 1 dependent stage steps, but every wave-instruction of it works for four instances and two of three
//     steps take P_{k+1} out of the lane's own registers (one DPP shift per lane boundary);
//   * vector sweeps: the lane composes its three affine stage maps, a Hillis-Steele scan over the row (4 steps of row_shr / row_shl
//     moves), then the lane applies its maps;
//   * reductions: butterfly all-reduce inside the row by DPP moves.  One wavefront per SIMD waits for every ds_bpermute round trip and
//     for every LDS read inside a lane-dependent branch on the spot: nothing of either kind is left in the interior-point loop;
//   * the rows: the corrector's right-hand side comes out of the predictor's row pass (no barrier pass), its multiplier steps are
//     kept for the update (no third pass over the rows' reciprocals);
//   * three stage objects are ~213 doubles per lane: r, q and the slack weights are parked in LDS columns, the rest fills 256 VGPRs +
//     ~245 AGPRs without scratch;
//   * the sensitivity pass (dV/dp, du0*/dp) runs at the end of the same kernel on the iterate in its registers (sens_tail).
//
// The iteration is the one of SmallSolver<LinearDev> (same constants, same formulas per stage and row; the sums of the
// reductions and of the scans associate differently, and the predictor's ratio test uses (t + dt) / t for -dlam / lam), so statuses
// and iteration counts equal the oracle port's up to stopping tests met within rounding.  Everything mpcrl_solve offers for the model
// is here: stored / cold iterates, the per-instance cold mask, MPCRL_COLD_DUAL, Q-mode, per-instance parameters, RTI, the divergence
// exit, general box bounds with the L1-soft first state, the Lagrangian, both sensitivities.  DESIGN.md §3.1 has the measurements.
#pragma once
#include "small_kernel.hpp"

namespace mpcrl {

// ---- cross-lane moves inside a DPP row (16 lanes): VALU moves, no LDS-crossbar round trip.  All 64 lanes must be active.
template <int CTRL>
MPCRL_DI double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// butterfly all-reduce over a row (RL = 16) or a half row (RL = 8): partners lane ^ 1, lane ^ 2 (quad permutations), 7 - lane of the
// half row, 15 - lane of the row.  Every lane ends with the same bits (the additions commute), no broadcast needed.
template <int NMAX, int NSUM, int RL>
MPCRL_DI void row_reduce(double *mx, double *sm) {
#define MPCRL_ROW_STEP(CTRL)                                                                  \
    {                                                                                         \
        double om[NMAX > 0 ? NMAX : 1], os[NSUM > 0 ? NSUM : 1];                              \
        _Pragma("unroll") for (int i = 0; i < NMAX; ++i) om[i] = dpp_move<CTRL>(mx[i]);       \
        _Pragma("unroll") for (int i = 0; i < NSUM; ++i) os[i] = dpp_move<CTRL>(sm[i]);       \
        _Pragma("unroll") for (int i = 0; i < NMAX; ++i) mx[i] = fmax(mx[i], om[i]);          \
        _Pragma("unroll") for (int i = 0; i < NSUM; ++i) sm[i] += os[i];                      \
    }
    MPCRL_ROW_STEP(0xB1) MPCRL_ROW_STEP(0x4E) MPCRL_ROW_STEP(0x141)
    if constexpr (RL == 16) MPCRL_ROW_STEP(0x140)
#undef MPCRL_ROW_STEP
}

// RL = 16 / 8: the instance owns one DPP row / half row (the live lanes first): reductions and scans by DPP moves.  (Round 5 also had
// RL = 0, packed segments of any length with __shfl, for horizons whose three-stage lanes outgrow a row; round 6 runs those with four
// stages per lane in rows — lq_solve_kernel<4, 16> — and the packed form is gone.)
template <int SPL, int RL>
struct LqSolver {
    static_assert(RL == 8 || RL == 16, "an instance owns a DPP half row or row");
    static constexpr int NX = 2, NU = 1, NW = 3;
    const SmallSpec &sp;
    const int N, lpi, lpl, pos, base;   // lpi: lanes of the slot of the instance; lpl: the ones that hold live stages
    bool qmode;
    const double *th;     // LDS: the instance's 12 parameters (A column-major 4, B 2, b 2, V_0, f 3)
    const double *bt;     // LDS: bounds by stage kind (first / interior / terminal): [kind][lb 3 | ub 3]
    // per stage of the lane
    int kst[SPL], kind[SPL];
    bool first[SPL], term[SPL], dead[SPL], softs[SPL];
    unsigned hasm[SPL];
    double ck[SPL];
    double x[SPL][NX], u[SPL], nu[SPL][NX];
    // parked in LDS ([slot][lane], the lane's own column: no synchronisation): what is read once per QP or once per row pass — the
    // dynamics defect r, the cost gradient q, the weights of the L1 slacks.  24 doubles of a lane's ~220: the kernel spills without this.
    double *park;
    static constexpr int PK_R = 0, PK_Q = PK_R + SPL * NX, PK_ZW = PK_Q + SPL * NW, PK_N = PK_ZW + SPL * 2;
    MPCRL_DI double r_(int j, int i) const { return park[(PK_R + j * NX + i) * 64]; }
    MPCRL_DI double q_(int j, int i) const { return park[(PK_Q + j * NW + i) * 64]; }
    double lam[SPL][2][NW], t[SPL][2][NW], aff[SPL][2][NW];
    double s[SPL][2], lams[SPL][2], ts[SPL][2], affs[SPL][2];
    double dx[SPL][NX], du[SPL], nuq[SPL][NX], Dx[SPL][NX], Du[SPL], Dnu[SPL][NX];
    double K[SPL][NX], Li[SPL], kff[SPL], P[SPL][3], p[SPL][NX];
    double rg[SPL][NW], rb[SPL][NX], rt[SPL][NW], Dg[SPL][NW];
    double n_rows_c = -1.0;
    double x0r[NX], u0r;
#ifdef MPCRL_PROFILE_PHASES
    unsigned long long phw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pht = 0;
#endif

    MPCRL_DI LqSolver(const SmallSpec &sp_, int lpi_, int lpl_, int pos_, int base_) : sp(sp_), N(sp_.N), lpi(lpi_), lpl(lpl_), pos(pos_), base(base_) {}

    // reductions over the lanes of the instance, result in all of them
    template <int NMAX, int NSUM>
    MPCRL_DI void red(double *mx, double *sm) const {
        row_reduce<NMAX, NSUM, RL>(mx, sm);
    }
    MPCRL_DI double red_sum(double v) const {
        red<0, 1>(nullptr, &v);
        return v;
    }
    MPCRL_DI double red_max(double v) const {   // (of non-negative values)
        red<1, 0>(&v, nullptr);
        return v;
    }

    MPCRL_DI static constexpr int sym(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
    MPCRL_DI double A_(int i, int j) const { return th[j * 2 + i]; }      // column-major in p (linear_system/acados.py:60-62,89-90)
    MPCRL_DI double B_(int i) const { return th[4 + i]; }
    MPCRL_DI double BA(int m, int i) const { return i < NU ? B_(m) : A_(m, i - NU); }
    MPCRL_DI double lbv(int j, int i) const { return bt[kind[j] * 6 + i]; }
    MPCRL_DI double ubv(int j, int i) const { return bt[kind[j] * 6 + 3 + i]; }
    MPCRL_DI bool has(int j, int sd, int i) const { return (hasm[j] >> (2 * i + sd)) & 1u; }
    MPCRL_DI bool softc(int j, int i) const { return i == NU && softs[j]; }
    MPCRL_DI double zw(int j, int sd) const { return park[(PK_ZW + j * 2 + sd) * 64]; }
    MPCRL_DI bool fixed(int j, int i) const { return first[j] && (i >= NU || qmode); }
    MPCRL_DI double vc(int j, int i) const { return i < NU ? u[j] : x[j][i - NU]; }
    MPCRL_DI double dvq(int j, int i) const { return i < NU ? (term[j] ? 0.0 : du[j]) : dx[j][i - NU]; }
    MPCRL_DI double Dvq(int j, int i) const { return i < NU ? (term[j] ? 0.0 : Du[j]) : Dx[j][i - NU]; }
    // the row passes of the interior-point iteration fetch the six bounds of a stage at once, ahead of the lane-dependent branches
    // around its rows (a read inside a branch is waited for on the spot: one wavefront per SIMD has nothing to hide it behind)
    double cbl[NW], cbu[NW], czw[2];      // (and the two slack weights of the stage)
    MPCRL_DI void load_bounds(int j) {
#pragma unroll
        for (int i = 0; i < NW; ++i) cbl[i] = lbv(j, i), cbu[i] = ubv(j, i);
        czw[0] = zw(j, 0), czw[1] = zw(j, 1);
    }
    MPCRL_DI double bslack_c(int j, int sd, int i, double v) const {
        const double sv = softc(j, i) ? s[j][sd] : 0.0;
        return sd ? cbu[i] - v + sv : v + sv - cbl[i];
    }
    MPCRL_DI double bslack(int j, int sd, int i, double v) const {
        const double sv = softc(j, i) ? s[j][sd] : 0.0;
        return sd ? ubv(j, i) - v + sv : v + sv - lbv(j, i);
    }
    // Hessian of the unscaled stage cost between coordinates of v = [u; x]: the identity, the terminal stage the symmetrised P of the model
    MPCRL_DI double Hs(int j, int a, int b) const {
        if (!term[j]) return a == b ? 1.0 : 0.0;
        if (a < NU || b < NU) return 0.0;
        return 0.5 * (sp.consts[(a - NU) * 2 + (b - NU)] + sp.consts[(b - NU) * 2 + (a - NU)]);
    }

    // stage set-up: kinds, cost scalings, bound rows
    MPCRL_DI void init_stages() {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int k = SPL * pos + j;
            kst[j] = k, first[j] = k == 0, term[j] = k >= N, dead[j] = k > N;
            kind[j] = k == 0 ? 0 : (k >= N ? 2 : 1);
            if (sp.cost_kind == 0)
                ck[j] = k >= N ? 1.0 : sp.dT;                                                       // nlp.py:1044-1055
            else
                ck[j] = k == 0 ? sp.dT : (k >= N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);   // nlp.py:1083-1091
            if (dead[j]) ck[j] = 0.0;
            unsigned h = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) h |= (lbv(j, i) > -NO_BOUND ? 1u : 0u) << (2 * i) | (ubv(j, i) < NO_BOUND ? 1u : 0u) << (2 * i + 1);
            hasm[j] = dead[j] ? 0u : h;
            softs[j] = k > 0 && k < N && sp.soft[NU] != 0;
            const double g = sp.dT * pow(sp.gamma, (double)k);
            park[(PK_ZW + j * 2) * 64] = sp.zl[NU] * g, park[(PK_ZW + j * 2 + 1) * 64] = sp.zu[NU] * g;
        }
    }

    // ---- linearise: r = F(x, u) - x_next, q = c_k grad l; returns c_k l_k (+ slack penalties)
    MPCRL_DI double linearize(int j, const double *xn) {
        const double f0 = th[0] * x[j][0] + th[2] * x[j][1] + th[4] * u[j] + th[6], f1 = th[1] * x[j][0] + th[3] * x[j][1] + th[5] * u[j] + th[7];
        park[(PK_R + j * NX) * 64] = term[j] ? 0.0 : f0 - xn[0], park[(PK_R + j * NX + 1) * 64] = term[j] ? 0.0 : f1 - xn[1];
        double val, q[NW];
        if (!term[j]) {      // l = 1/2 y'y + f'y (+ V_0 at k = 0), y = [x; u]
            q[0] = u[j] + th[11], q[1] = x[j][0] + th[9], q[2] = x[j][1] + th[10];
            val = 0.5 * (x[j][0] * x[j][0] + x[j][1] * x[j][1] + u[j] * u[j]) + th[9] * x[j][0] + th[10] * x[j][1] + th[11] * u[j];
            if (first[j]) val += th[8];
        } else {
            const double p00 = sp.consts[0], p01 = 0.5 * (sp.consts[1] + sp.consts[2]), p11 = sp.consts[3];
            q[0] = 0.0, q[1] = p00 * x[j][0] + p01 * x[j][1], q[2] = p01 * x[j][0] + p11 * x[j][1];
            val = 0.5 * (x[j][0] * q[1] + x[j][1] * q[2]);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) park[(PK_Q + j * NW + i) * 64] = q[i] * ck[j];
        val *= ck[j];
        if (softs[j]) val += zw(j, 0) * s[j][0] + zw(j, 1) * s[j][1];
        return val;
    }
    MPCRL_DI double GTnu(int j, const double *nun, const double *nuo, int i) const {
        double a = 0.0;
        if (!term[j]) {
#pragma unroll
            for (int m = 0; m < NX; ++m) a = fma(BA(m, i), nun[m], a);
        }
        if (i >= NU && !first[j]) a -= nuo[i - NU];
        return a;
    }
    MPCRL_DI void nlp_res_local(int j, const double *nun, double *res) const {
        double rs = 0, re = 0, ri = 0, rc = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (term[j] && i < NU) continue;
            if (!fixed(j, i)) {
                double g = q_(j, i) + GTnu(j, nun, nu[j], i);
                if (has(j, 0, i)) g -= lam[j][0][i];
                if (has(j, 1, i)) g += lam[j][1][i];
                rs = fmax(rs, fabs(g));
            }
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
                if (has(j, sd, i)) {
                    const double h = -bslack(j, sd, i, vc(j, i));
                    ri = fmax(ri, h);
                    rc = fmax(rc, fabs(lam[j][sd][i] * h));
                    if (softc(j, i)) {
                        ri = fmax(ri, -s[j][sd]);
                        rc = fmax(rc, fabs(lams[j][sd] * s[j][sd]));
                        rs = fmax(rs, fabs(zw(j, sd) - lam[j][sd][i] - lams[j][sd]));
                    }
                }
        }
        if (!term[j]) re = fmax(fabs(r_(j, 0)), fabs(r_(j, 1)));
        if (first[j]) {
            re = fmax(re, fmax(fabs(x[j][0] - x0r[0]), fabs(x[j][1] - x0r[1])));
            if (qmode) re = fmax(re, fabs(u[j] - u0r));
        }
        if (dead[j]) rs = re = ri = rc = 0.0;
        res[0] = fmax(res[0], rs), res[1] = fmax(res[1], re), res[2] = fmax(res[2], ri), res[3] = fmax(res[3], rc);
    }

    // ---- one backward Riccati stage (SmallSolver::riccati_stage<true> for nx = 2, nu = 1): (Pn, pn) of stage k + 1 -> K, Li, kff, P, p
    // ab: [A B] of the instance in registers (the serial loop must not wait for LDS inside its lane-dependent branches)
    MPCRL_DI bool riccati_stage(int j, const double *Pn, const double *pn, const double *ab) {
        auto BA = [&](int m, int i) { return i < NU ? ab[4 + m] : ab[(i - NU) * 2 + m]; };
        const double hsc = ck[j];
        if (term[j]) {
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) P[j][sym(a, b)] = fma(hsc, Hs(j, NU + a, NU + b), a == b ? Dg[j][NU + a] : 0.0);
            p[j][0] = rt[j][NU], p[j][1] = rt[j][NU + 1];
            return true;
        }
        bool ok = true;
        double cc[NX], mv[NW];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            double v = pn[a];
#pragma unroll
            for (int b = 0; b < NX; ++b) v = fma(Pn[sym(a, b)], rb[j][b], v);
            cc[a] = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            double v = rt[j][i];
#pragma unroll
            for (int m = 0; m < NX; ++m) v = fma(BA(m, i), cc[m], v);
            mv[i] = v;
        }
        double T[NX * NW], Mm[NW * (NW + 1) / 2];
#pragma unroll
        for (int a = 0; a < NX; ++a)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < NX; ++m) v = fma(Pn[sym(a, m)], BA(m, i), v);
                T[a * NW + i] = v;
            }
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double v = fma(hsc, Hs(j, a, b), a == b ? Dg[j][a] : 0.0);
#pragma unroll
                for (int m = 0; m < NX; ++m) v = fma(BA(m, a), T[m * NW + b], v);
                Mm[sym(a, b)] = v;
            }
        if (first[j] && qmode) {
            K[j][0] = K[j][1] = 0.0, Li[j] = 0.0;
        } else {
            ok = Mm[0] > 0.0;
            Li[j] = fast_rcp(Mm[0]);   // scalar pivot: 1 / R itself
            K[j][0] = Mm[sym(NU, 0)] * Li[j], K[j][1] = Mm[sym(NU + 1, 0)] * Li[j];
        }
#pragma unroll
        for (int a = 0; a < NX; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) P[j][sym(a, b)] = Mm[sym(NU + a, NU + b)] - Mm[sym(NU + a, 0)] * K[j][b];
        kff[j] = (first[j] && qmode) ? 0.0 : mv[0] * Li[j];
        p[j][0] = mv[NU] - K[j][0] * mv[0], p[j][1] = mv[NU + 1] - K[j][1] * mv[0];
        return ok;
    }

    // ---- factor sweep: serial over the lanes (wave-uniform loop), the three stages of a lane in sequence.  Every lane executes every
    // step: a lane whose turn has passed recomputes its stages from final neighbours to the same bits, one whose turn has not come
    // computes throw-away values its own turn overwrites.  Only the lane whose turn it is reports its pivots.
    MPCRL_DI bool factor() {
        bool ok = true;
        double ab[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) ab[e] = th[e];
        for (int ps = lpl - 1; ps >= 0; --ps) {
            double Pin[3], pin[NX];
#pragma unroll
            for (int i = 0; i < 3; ++i) Pin[i] = lane_dn(P[0][i]);
#pragma unroll
            for (int i = 0; i < NX; ++i) pin[i] = lane_dn(p[0][i]);
            bool okl = true;
#pragma unroll
            for (int j = SPL - 1; j >= 0; --j) {
                const bool o = j == SPL - 1 ? riccati_stage(j, Pin, pin, ab) : riccati_stage(j, P[j + 1 < SPL ? j + 1 : 0], p[j + 1 < SPL ? j + 1 : 0], ab);
                okl = okl && o;
            }
            ok = ok && (okl || pos != ps);
        }
        return ok;
    }

    // ---- affine maps y = M x + v of the vector recursions (2 x 2): composition and application
    struct Aff {
        double M[4], v[2];
    };
    MPCRL_DI static Aff compose(const Aff &outer, const Aff &inner) {   // outer o inner
        Aff o;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            o.v[a] = fma(outer.M[a * 2], inner.v[0], fma(outer.M[a * 2 + 1], inner.v[1], outer.v[a]));
#pragma unroll
            for (int b = 0; b < 2; ++b) o.M[a * 2 + b] = fma(outer.M[a * 2], inner.M[b], outer.M[a * 2 + 1] * inner.M[2 + b]);
        }
        return o;
    }
    MPCRL_DI static void apply(const Aff &m, const double *xin, double *y) {
        y[0] = fma(m.M[0], xin[0], fma(m.M[1], xin[1], m.v[0])), y[1] = fma(m.M[2], xin[0], fma(m.M[3], xin[1], m.v[1]));
    }
    // Hillis-Steele scan over the lanes of the instance: afterwards lane `pos` holds the composition of the lane maps from itself to
    // the end of the instance (down = true: partners at pos + s, own map OUTER) / from the start to itself (partners at pos - s)
    template <int SFT>
    MPCRL_DI void scan_step(Aff &m, bool down) const {
        Aff o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.M[i] = down ? dpp_move<0x100 + SFT>(m.M[i]) : dpp_move<0x110 + SFT>(m.M[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) o.v[i] = down ? dpp_move<0x100 + SFT>(m.v[i]) : dpp_move<0x110 + SFT>(m.v[i]);
        const bool valid = down ? pos + SFT < RL : pos - SFT >= 0;
        const Aff n = compose(m, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) m.M[i] = valid ? n.M[i] : m.M[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) m.v[i] = valid ? n.v[i] : m.v[i];
    }
    MPCRL_DI void scan(Aff &m, bool down) const {
        // row shifts (the lanes behind the live ones hold identity / constant maps)
        scan_step<1>(m, down), scan_step<2>(m, down), scan_step<4>(m, down);
        if constexpr (RL == 16) scan_step<8>(m, down);
    }

    // ---- vector-only backward sweep on the stored factors (SmallSolver::backward_scan): p_k, kff_k for the right-hand side rt
    MPCRL_DI void backward_vec() {
        double ab[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) ab[e] = th[e];
        auto A_ = [&](int i, int j) { return ab[j * 2 + i]; };
        auto B_ = [&](int i) { return ab[4 + i]; };
        // hb_k = P_k bb_{k-1}
        double hb[SPL][NX], bprev[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) bprev[i] = lane_up(rb[SPL - 1][i]);
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *bp = j == 0 ? bprev : rb[j > 0 ? j - 1 : 0];
#pragma unroll
            for (int a = 0; a < NX; ++a) hb[j][a] = fma(P[j][sym(a, 0)], bp[0], P[j][sym(a, 1)] * bp[1]);
        }
        double hbin[NX];      // hb of the next lane's first stage
#pragma unroll
        for (int i = 0; i < NX; ++i) hbin[i] = lane_dn(hb[0][i]);
        Aff sm[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *hn = j + 1 < SPL ? hb[j + 1 < SPL ? j + 1 : 0] : hbin;
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                double c = fma(-K[j][a], rt[j][0], rt[j][NU + a]);
#pragma unroll
                for (int m = 0; m < NX; ++m) {
                    const double acl = fma(-B_(m), K[j][a], A_(m, a));      // (Acl')(a, m)
                    sm[j].M[a * 2 + m] = term[j] ? 0.0 : acl;
                    c = term[j] ? c : fma(acl, hn[m], c);
                }
                sm[j].v[a] = term[j] ? rt[j][NU + a] : c;
            }
        }
        // lane map: first stage outermost
        Aff lm = sm[SPL - 1];
#pragma unroll
        for (int j = SPL - 2; j >= 0; --j) lm = compose(sm[j], lm);
        scan(lm, true);
        // p of the NEXT lane's first stage = its composed map applied to anything (the terminal map is constant): shift it down
        const double zero[NX] = {0.0, 0.0};
        double pfirst[NX], pin[NX];
        apply(lm, zero, pfirst);
#pragma unroll
        for (int i = 0; i < NX; ++i) pin[i] = lane_dn(pfirst[i]);
#pragma unroll
        for (int j = SPL - 1; j >= 0; --j) {      // (a terminal stage takes its constant: what follows it belongs to another instance)
            double y[NX];
            apply(sm[j], j == SPL - 1 ? pin : p[j + 1 < SPL ? j + 1 : 0], y);
            p[j][0] = term[j] ? sm[j].v[0] : y[0], p[j][1] = term[j] ? sm[j].v[1] : y[1];
        }
        // feed-forward: kff = (g_u + B'(p_{k+1} + hb_{k+1})) / R
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *pn = j + 1 < SPL ? p[j + 1 < SPL ? j + 1 : 0] : pin, *hn = j + 1 < SPL ? hb[j + 1 < SPL ? j + 1 : 0] : hbin;
            double mvu = rt[j][0];
#pragma unroll
            for (int m = 0; m < NX; ++m) mvu = fma(B_(m), pn[m] + hn[m], mvu);
            if (!term[j]) kff[j] = (first[j] && qmode) ? 0.0 : mvu * Li[j];
        }
    }

    // ---- forward sweep (SmallSolver::forward_scan): Dx, Du, Dnu
    MPCRL_DI void forward_vec() {
        double ab[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) ab[e] = th[e];
        auto A_ = [&](int i, int j) { return ab[j * 2 + i]; };
        auto B_ = [&](int i) { return ab[4 + i]; };
        Aff sm[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j)
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                sm[j].v[a] = term[j] ? 0.0 : fma(-B_(a), kff[j], rb[j][a]);
#pragma unroll
                for (int b = 0; b < NX; ++b) sm[j].M[a * 2 + b] = term[j] ? (a == b ? 1.0 : 0.0) : fma(-B_(a), K[j][b], A_(a, b));
            }
        Aff lm = sm[0];
#pragma unroll
        for (int j = 1; j < SPL; ++j) lm = compose(sm[j], lm);
        scan(lm, false);
        // lane `pos` now maps Dx_0 = 0 to the state AFTER its last stage: the next lane's first Dx
        const double zero[NX] = {0.0, 0.0};
        double xout[NX], xin[NX];
        apply(lm, zero, xout);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double up = lane_up(xout[i]);      // (the DPP move outside the select: under a lane mask it would read disabled lanes)
            xin[i] = pos == 0 ? 0.0 : up;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *xi = j == 0 ? xin : nullptr;
            if (j == 0)
                Dx[0][0] = xi[0], Dx[0][1] = xi[1];
            else
                apply(sm[j - 1], Dx[j - 1], Dx[j]);
            Du[j] = -fma(K[j][0], Dx[j][0], fma(K[j][1], Dx[j][1], kff[j]));
#pragma unroll
            for (int a = 0; a < NX; ++a) Dnu[j][a] = first[j] ? 0.0 : fma(P[j][sym(a, 0)], Dx[j][0], fma(P[j][sym(a, 1)], Dx[j][1], p[j][a]));
        }
    }

    // ---- interior point: per-row Newton quantities (SmallSolver::barrier_terms / row_steps)
    MPCRL_DI static double rm_(double l, double tt, double af, int pass, double smu) { return fma(l, tt, pass ? af - smu : 0.0); }
    MPCRL_DI void barrier_terms(int j, int i, double v, int pass, double smu, double &dg, double &er) const {
        dg = 0.0, er = 0.0;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            if (!has(j, sd, i)) continue;
            const double sg = sd ? -1.0 : 1.0;
            const double l1 = lam[j][sd][i], t1 = t[j][sd][i], it1 = fast_rcp(t1);
            const double w1 = l1 * it1;
            const double rd1 = t1 - bslack_c(j, sd, i, v);
            const double e1 = (rm_(l1, t1, aff[j][sd][i], pass, smu) - l1 * rd1) * it1;
            if (softc(j, i)) {
                const double l2 = lams[j][sd], t2 = ts[j][sd], it2 = fast_rcp(t2);
                const double w2 = l2 * it2;
                const double e2 = (rm_(l2, t2, affs[j][sd], pass, smu) - l2 * (t2 - s[j][sd])) * it2;
                const double rgs = czw[sd] - l1 - l2;
                const double iw = fast_rcp(w1 + w2);
                dg += w1 * w2 * iw;
                er += sg * (e1 * w2 - w1 * (rgs + e2)) * iw;
            } else {
                dg += w1;
                er += sg * e1;
            }
        }
    }
    // ca, ci (pass 0): the corrector's barrier term of the coordinate is the predictor's + sum_rows (ca - smu ci) — its complementarity
    // targets differ from the predictor's by aff - smu, and the term is linear in them — so the corrector needs no pass over the rows
    // for its right-hand side: ca = sg aff / t, ci = sg / t (hard row); the soft pair eliminates its slack first
    MPCRL_DI void row_steps(int j, int i, int sd, double v, double dv, int pass, double smu, double &dt1, double &dl1, double &dt2, double &dl2,
                            double &dss, double &rat, double *ca = nullptr, double *ci = nullptr) const {
        const double sg = sd ? -1.0 : 1.0;
        const double l1 = lam[j][sd][i], t1 = t[j][sd][i], it1 = fast_rcp(t1);
        const double rd1 = t1 - bslack_c(j, sd, i, v);
        const double rm1 = rm_(l1, t1, aff[j][sd][i], pass, smu);
        dss = 0.0, dt2 = 0.0, dl2 = 0.0, rat = 0.0;
        if (softc(j, i)) {
            const double l2 = lams[j][sd], t2 = ts[j][sd], it2 = fast_rcp(t2);
            const double w1 = l1 * it1, w2 = l2 * it2;
            const double rd2 = t2 - s[j][sd];
            const double rm2 = rm_(l2, t2, affs[j][sd], pass, smu);
            const double e1 = (rm1 - l1 * rd1) * it1, e2 = (rm2 - l2 * rd2) * it2;
            const double rgs = czw[sd] - l1 - l2;
            dss = -(rgs + e1 + e2 + sg * w1 * dv) * fast_rcp(w1 + w2);
            dt2 = -rd2 + dss;
            dl2 = (-rm2 - l2 * dt2) * it2;
            // (predictor: r_m = lam t, so -dlam / lam = (t + dt) / t — no reciprocal of the multiplier)
            rat = fmax(pass ? -dl2 * fast_rcp(l2) : (t2 + dt2) * it2, -dt2 * it2);
            dt1 = -rd1 + sg * dv + dss;
            dl1 = (-rm1 - l1 * dt1) * it1;
            if (ca) {
                const double iw = fast_rcp(w1 + w2), c1 = sg * it1 * w2 * iw, c2 = sg * w1 * it2 * iw;
                *ca = c1 * (dl1 * dt1) - c2 * (dl2 * dt2), *ci = c1 - c2;
            }
        } else {
            dt1 = -rd1 + sg * dv;
            dl1 = (-rm1 - l1 * dt1) * it1;
            if (ca) *ci = sg * it1, *ca = *ci * (dl1 * dt1);
        }
        rat = fmax(rat, fmax(pass ? -dl1 * fast_rcp(l1) : (t1 + dt1) * it1, -dt1 * it1));
    }

    // ---- sensitivities at the returned iterate (SmallSolver::sensitivities for this model, nlp.py:1211,1399-1424), run by the solve
    // kernel itself at the end of a wavefront's life — the iterate is in its registers and the factor sweep is at hand, so the
    // second launch of the one-stage layout (small_sens_kernel: 0.05 ms per 4096 instances, a tenth of the step) is not needed.
    // dV/dp = dL/dp: sum_k nu_{k+1}' dF/dp + c_k dl/dp.  du0*/dp: ONE adjoint solve K y = -e_{u0} with the exact Lagrangian Hessian
    // (the map is linear: c_k hess l) and the barrier diagonal lam / t of the bound rows (slacks are constants of the mirror,
    // quirk q1), then du0*/dp = -sum_k (nu_{k+1}' d2F/dp dv y_v + y_{nu,k+1}' dF/dp + c_k y_v' d2l/dv dp).  F = A x + B u + b, p = [A
    // column-major, B, b, V_0, f]: every derivative is written out.
    MPCRL_DI void sens_tail(const SmallArgs &a, long inst, bool valid, int status) {
        constexpr int NP = 12;
        const bool sv = valid && (status == 0 || status == 2);
        double *dVa = a.dV ? a.dV + inst * NP : nullptr, *dpia = a.dpi ? a.dpi + inst * NU * NP : nullptr;
        const bool want_v = (a.flags & 1) && dVa, want_pi = (a.flags & 2) && dpia && !qmode;   // wave-uniform
        double nuin[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) nuin[i] = lane_dn(nu[0][i]);
        if (want_v) {
            double acc[NP];
#pragma unroll
            for (int d = 0; d < NP; ++d) acc[d] = 0.0;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const double *nun = j + 1 < SPL ? nu[j + 1 < SPL ? j + 1 : 0] : nuin;
                const double w = term[j] ? 0.0 : 1.0, c = term[j] ? 0.0 : ck[j];
                acc[0] = fma(w * nun[0], x[j][0], acc[0]), acc[1] = fma(w * nun[1], x[j][0], acc[1]);
                acc[2] = fma(w * nun[0], x[j][1], acc[2]), acc[3] = fma(w * nun[1], x[j][1], acc[3]);
                acc[4] = fma(w * nun[0], u[j], acc[4]), acc[5] = fma(w * nun[1], u[j], acc[5]);
                acc[6] += w * nun[0], acc[7] += w * nun[1];
                acc[8] += first[j] ? c : 0.0;
                acc[9] = fma(c, x[j][0], acc[9]), acc[10] = fma(c, x[j][1], acc[10]), acc[11] = fma(c, u[j], acc[11]);
            }
            red<0, NP>(nullptr, acc);
            if (pos == 0 && valid)
#pragma unroll
                for (int d = 0; d < NP; ++d) dVa[d] = sv ? acc[d] : 0.0;
        } else if (dVa && pos == 0 && valid) {
#pragma unroll
            for (int d = 0; d < NP; ++d) dVa[d] = 0.0;
        }
        if (!want_pi) {
            if (dpia && pos == 0 && valid)
#pragma unroll
                for (int d = 0; d < NU * NP; ++d) dpia[d] = 0.0;
            return;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                double d = 0.0;
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (has(j, sd, i)) d += fmin(lam[j][sd][i] / t[j][sd][i], SENS_W_MAX);
                Dg[j][i] = d;
                rt[j][i] = (first[j] && i == 0) ? -1.0 : 0.0;
            }
            rb[j][0] = rb[j][1] = 0.0;
        }
        const bool okf = factor();
        forward_vec();
        const bool okall = red_max(okf ? 0.0 : 1.0) < 0.5;
        double ynin[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) ynin[i] = lane_dn(Dnu[0][i]);
        double acc[NP];
#pragma unroll
        for (int d = 0; d < NP; ++d) acc[d] = 0.0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *nun = j + 1 < SPL ? nu[j + 1 < SPL ? j + 1 : 0] : nuin, *ynn = j + 1 < SPL ? Dnu[j + 1 < SPL ? j + 1 : 0] : ynin;
            const double w = term[j] ? 0.0 : 1.0, c = term[j] ? 0.0 : ck[j], yu = term[j] ? 0.0 : Du[j];
            const double n0 = w * nun[0], n1 = w * nun[1], y0 = w * ynn[0], y1 = w * ynn[1];
            acc[0] = fma(n0, Dx[j][0], fma(y0, x[j][0], acc[0])), acc[1] = fma(n1, Dx[j][0], fma(y1, x[j][0], acc[1]));
            acc[2] = fma(n0, Dx[j][1], fma(y0, x[j][1], acc[2])), acc[3] = fma(n1, Dx[j][1], fma(y1, x[j][1], acc[3]));
            acc[4] = fma(n0, yu, fma(y0, u[j], acc[4])), acc[5] = fma(n1, yu, fma(y1, u[j], acc[5]));
            acc[6] += y0, acc[7] += y1;
            acc[9] = fma(c, Dx[j][0], acc[9]), acc[10] = fma(c, Dx[j][1], acc[10]), acc[11] = fma(c, yu, acc[11]);
        }
        red<0, NP>(nullptr, acc);
        if (pos == 0 && valid)
#pragma unroll
            for (int d = 0; d < NP; ++d) dpia[d] = sv ? (okall ? -acc[d] : NAN) : 0.0;
    }

    // ---- Mehrotra predictor-corrector on the QP of the current linearisation (SmallSolver::qp_solve, three stages per lane)
    MPCRL_DI bool qp_solve(bool act, int &n_it, double warm_mu, double tol_res, double tol_mu) {
        const bool warm = warm_mu > 0.0;
        if (act) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
#pragma unroll
                for (int i = 0; i < NX; ++i) dx[j][i] = first[j] ? x0r[i] - x[j][i] : 0.0, nuq[j][i] = warm ? nu[j][i] : 0.0;
                du[j] = (first[j] && qmode) ? u0r - u[j] : 0.0;
            }
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
        for (int j = 0; j < SPL; ++j)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                if (term[j] && i < NU) continue;
                const double v = vc(j, i) + dvq(j, i);
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    if (!has(j, sd, i)) continue;
                    cnt += 1.0;
                    if (softc(j, i)) {
                        cnt += 1.0;
                        if (act) {
                            if (warm) {
                                double l = lams[j][sd], tt = fmax(s[j][sd], ts[j][sd]);
                                recentre(l, tt);
                                lams[j][sd] = l, ts[j][sd] = tt;
                            } else
                                s[j][sd] = 0.0, ts[j][sd] = IPM_T_MIN, lams[j][sd] = IPM_MU0 / IPM_T_MIN;
                        }
                    }
                    if (act) {
                        const double sl = bslack(j, sd, i, v);
                        if (warm) {
                            double l = lam[j][sd][i], tt = fmax(sl, t[j][sd][i]);
                            recentre(l, tt);
                            lam[j][sd][i] = l, t[j][sd][i] = tt;
                        } else {
                            t[j][sd][i] = fmax(sl, IPM_T_MIN);
                            lam[j][sd][i] = IPM_MU0 * fast_rcp(t[j][sd][i]);
                        }
                    }
                }
            }
        if (n_rows_c < 0.0) n_rows_c = red_sum(cnt);
        const double n_rows = n_rows_c;
        PHW(10);
        bool qlive = act, ok = false;
        double rinf_c = 0.0, musum_c = 0.0;
        for (int it = 0;; ++it) {
            double rinf, musum;
            if (it == 0) {
                // residuals of the linear equations, once per QP (later iterations scale them)
                double dxin[NX], nuqin[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) dxin[i] = lane_dn(dx[0][i]), nuqin[i] = lane_dn(nuq[0][i]);
                double rloc = 0.0, muloc = 0.0;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                    const double *dxn = j + 1 < SPL ? dx[j + 1 < SPL ? j + 1 : 0] : dxin, *nuqn = j + 1 < SPL ? nuq[j + 1 < SPL ? j + 1 : 0] : nuqin;
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        double a = 0.0;
                        if (!term[j]) {
                            a = r_(j, i) - dxn[i];
#pragma unroll
                            for (int b = 0; b < NX; ++b) a = fma(A_(i, b), dx[j][b], a);
                            a = fma(B_(i), du[j], a);
                        }
                        rb[j][i] = a;
                        rloc = fmax(rloc, fabs(a));
                    }
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        rg[j][i] = 0.0;
                        if (term[j] && i < NU) continue;
                        double a = q_(j, i) + GTnu(j, nuqn, nuq[j], i), hdv = 0.0;
#pragma unroll
                        for (int b = 0; b < NW; ++b) hdv = fma(Hs(j, i, b), dvq(j, b), hdv);
                        a = fma(ck[j], hdv, a);
                        if (has(j, 0, i)) a -= lam[j][0][i];
                        if (has(j, 1, i)) a += lam[j][1][i];
                        if (fixed(j, i) || dead[j]) a = 0.0;
                        rg[j][i] = a;
                        rloc = fmax(rloc, fabs(a));
                        const double v = vc(j, i) + dvq(j, i);
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd) {
                            if (!has(j, sd, i)) continue;
                            rloc = fmax(rloc, fabs(t[j][sd][i] - bslack(j, sd, i, v)));
                            muloc = fma(lam[j][sd][i], t[j][sd][i], muloc);
                            if (softc(j, i)) {
                                rloc = fmax(rloc, fabs(ts[j][sd] - s[j][sd]));
                                rloc = fmax(rloc, fabs(zw(j, sd) - lam[j][sd][i] - lams[j][sd]));
                                muloc = fma(lams[j][sd], ts[j][sd], muloc);
                            }
                        }
                    }
                }
                red<1, 1>(&rloc, &muloc);
                rinf = rloc, musum = muloc;
            } else
                rinf = rinf_c, musum = musum_c;
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
#pragma unroll
            for (int j = 0; j < SPL; ++j)
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (i == 0) load_bounds(j);
                    const double v = vc(j, i) + dvq(j, i);
                    double e;
                    barrier_terms(j, i, v, 0, 0.0, Dg[j][i], e);
                    rt[j][i] = rg[j][i] + e;
                }
            PHW(1);
            const bool okf = factor();
            PHW(2);
            forward_vec();
            PHW(3);
            double okbad = okf ? 0.0 : 1.0, rmax = 1.0, c12[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < SPL; ++j)
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (i == 0) load_bounds(j);
                    if (term[j] && i < NU) continue;
                    const double v = vc(j, i) + dvq(j, i), dv = Dvq(j, i);
                    Dg[j][i] = 0.0;      // (the factor is done with the barrier diagonal: the place collects ci, rt collects ca)
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        if (!has(j, sd, i)) continue;
                        double dt1, dl1, dt2, dl2, dss, rat, ca, ci;
                        row_steps(j, i, sd, v, dv, 0, 0.0, dt1, dl1, dt2, dl2, dss, rat, &ca, &ci);
                        rt[j][i] += ca, Dg[j][i] += ci;
                        rmax = fmax(rmax, rat);
                        aff[j][sd][i] = dl1 * dt1;
                        c12[0] = fma(lam[j][sd][i], dt1, fma(t[j][sd][i], dl1, c12[0])), c12[1] = fma(dl1, dt1, c12[1]);
                        if (softc(j, i)) {
                            affs[j][sd] = dl2 * dt2;
                            c12[0] = fma(lams[j][sd], dt2, fma(ts[j][sd], dl2, c12[0])), c12[1] = fma(dl2, dt2, c12[1]);
                        }
                    }
                }
            {
                double two[2] = {rmax, okbad};
                red<2, 2>(two, c12);
                rmax = two[0];
                if (two[1] > 0.5) qlive = false;   // non-positive pivot: QP failure
            }
            PHW(4);
            const double a_aff = fast_rcp(rmax);
            const double mu_aff = fma(a_aff, fma(a_aff, c12[1], c12[0]), musum) * inv_rows;
            const double ratio = mu > 0.0 ? mu_aff * fast_rcp(mu) : 0.0;
            const double smu = ratio * ratio * ratio * mu;
            const double frac = IPM_FRAC;   // (LQ model: fixed fraction to the boundary)
            // ---- corrector (same factorisation, vector sweeps only)
#pragma unroll
            for (int j = 0; j < SPL; ++j)
#pragma unroll
                for (int i = 0; i < NW; ++i) rt[j][i] = fma(-smu, Dg[j][i], rt[j][i]);
            PHW(5);
            backward_vec();
            PHW(6);
            forward_vec();
            PHW(7);
            rmax = 1.0;
            double d12[2] = {0.0, 0.0};
            // the rows' multiplier steps of the corrector are kept for the update below in the places of quantities that are dead by
            // now (dlam in aff / affs, ds in rt): the update recomputes only the cheap dt = -r_d + sg dv + ds, not the rows' reciprocals
#pragma unroll
            for (int j = 0; j < SPL; ++j)
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (i == 0) load_bounds(j);
                    if (term[j] && i < NU) continue;
                    const double v = vc(j, i) + dvq(j, i), dv = Dvq(j, i);
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        if (!has(j, sd, i)) continue;
                        double dt1, dl1, dt2, dl2, dss, rat;
                        row_steps(j, i, sd, v, dv, 1, smu, dt1, dl1, dt2, dl2, dss, rat);
                        rmax = fmax(rmax, rat);
                        aff[j][sd][i] = dl1;
                        d12[0] = fma(lam[j][sd][i], dt1, fma(t[j][sd][i], dl1, d12[0])), d12[1] = fma(dl1, dt1, d12[1]);
                        if (softc(j, i)) {
                            affs[j][sd] = dl2, rt[j][sd] = dss;
                            d12[0] = fma(lams[j][sd], dt2, fma(ts[j][sd], dl2, d12[0])), d12[1] = fma(dl2, dt2, d12[1]);
                        }
                    }
                }
            red<1, 2>(&rmax, d12);
            if (qlive) {
                const double alpha = fmin(1.0, frac * fast_rcp(rmax));
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        if (i == 0) load_bounds(j);
                        if (term[j] && i < NU) continue;
                        const double v = vc(j, i) + dvq(j, i), dv = Dvq(j, i);
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd) {
                            if (!has(j, sd, i)) continue;
                            const double sg = sd ? -1.0 : 1.0, rd1 = t[j][sd][i] - bslack_c(j, sd, i, v);
                            double dss = 0.0;
                            if (softc(j, i)) {
                                dss = rt[j][sd];
                                const double dt2 = -(ts[j][sd] - s[j][sd]) + dss;
                                lams[j][sd] = fma(alpha, affs[j][sd], lams[j][sd]);
                                ts[j][sd] = fma(alpha, dt2, ts[j][sd]);
                                s[j][sd] = fma(alpha, dss, s[j][sd]);
                            }
                            const double dt1 = -rd1 + sg * dv + dss;
                            lam[j][sd][i] = fma(alpha, aff[j][sd][i], lam[j][sd][i]);
                            t[j][sd][i] = fma(alpha, dt1, t[j][sd][i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NX; ++i) dx[j][i] = fma(alpha, Dx[j][i], dx[j][i]), nuq[j][i] = fma(alpha, Dnu[j][i], nuq[j][i]);
                    du[j] = fma(alpha, Du[j], du[j]);
                    const double om = 1.0 - alpha;
#pragma unroll
                    for (int i = 0; i < NX; ++i) rb[j][i] *= om;
#pragma unroll
                    for (int i = 0; i < NW; ++i) rg[j][i] *= om;
                }
                rinf_c = (1.0 - alpha) * rinf;
                musum_c = fma(alpha, fma(alpha, d12[1], d12[0]), musum);
            }
            PHW(8);
        }
        return ok;
    }
};

// =====================================================================================================
// kernel: floor(64 / ceil((N + 1) / SPL)) instances per wavefront (N = 40, SPL = 3: four)
// =====================================================================================================
template <int SPL>
__host__ __device__ constexpr int lq_lanes_per_instance(int N) { return (N + 1 + SPL - 1) / SPL; }

template <int SPL, int RL>
__global__ void __launch_bounds__(64, 1) lq_solve_kernel(const SmallSpec sp, const SmallArgs a) {
    constexpr int NX = 2, NU = 1, NW = 3, MAXI = 8;
#ifdef MPCRL_PROFILE_PHASES
    const unsigned long long rt0_ = wall_clock64();
#endif
    const int lane = threadIdx.x, N = sp.N, lpl = lq_lanes_per_instance<SPL>(N), lpi = RL, ipw = min(64 / lpi, MAXI);
    const int slot = lane / lpi, pos = lane - slot * lpi, base = slot * lpi;
    long inst = (long)blockIdx.x * ipw + slot;
    const bool valid = slot < ipw && inst < a.B;
    if (!valid) inst = a.B - 1;   // dead lanes shadow the last instance and never store
    if (a.perm) inst = a.perm[inst];
    __shared__ double th_lds[(MAXI + 1) * 12], bt_lds[18], park_lds[LqSolver<SPL, RL>::PK_N * 64];
    LqSolver<SPL, RL> S(sp, lpi, lpl, pos, base);
    S.park = park_lds + lane;
    S.qmode = a.u0fix != nullptr;
    {   // the instance's parameters and the bound table (one copy per wavefront)
        double *thw = th_lds + (slot < ipw ? slot : MAXI) * 12;
        const double *th = a.theta + (size_t)inst * a.theta_stride;
        // (the lanes past the last instance slot fill THEIR slot completely, however few they are: at N = 60..62 there is one)
        const int spare = 64 - ipw * lpi;
        for (int e = slot < ipw ? pos : lane - ipw * lpi; e < 12; e += slot < ipw ? lpi : spare) thw[e] = th[e];
        if (lane < 18) {
            const int kd = lane / 6, e = lane - kd * 6, sd = e / 3, i = e - sd * 3;
            double v;
            if (kd == 0)
                v = (i < NU && !S.qmode) ? (sd ? sp.ub0[0] : sp.lb0[0]) : (sd ? 1e30 : -1e30);
            else if (kd == 2)
                v = i >= NU ? (sd ? sp.ube[i >= NU ? i - NU : 0] : sp.lbe[i >= NU ? i - NU : 0]) : (sd ? 1e30 : -1e30);
            else
                v = sd ? sp.ub[i] : sp.lb[i];
            bt_lds[lane] = v;
        }
        S.th = thw, S.bt = bt_lds;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    S.init_stages();
    const double *x0 = a.x0 + inst * NX;
    S.x0r[0] = x0[0], S.x0r[1] = x0[1];
    S.u0r = S.qmode ? a.u0fix[inst * NU] : 0.0;
    // ---- iterate: stored (warm) or the reference's cold start (MPC.reset, mpc.py:204-210)
    const size_t nb = (size_t)(N + 1) * NW;
    const bool cold = (a.flags & 8) || (a.cold && a.cold[inst]);
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int k = S.kst[j], kc = k <= N ? k : N;
        const bool first = S.first[j], term = S.term[j], dead = S.dead[j];
        const double *bnd = a.BND + (size_t)inst * 10 * nb + (size_t)kc * NW;
        const bool load = !(a.flags & 8);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double xs = load ? a.X[(inst * (N + 1) + kc) * NX + i] : 0.0, ns = (load && !first) ? a.PI[(inst * N + kc - 1) * NX + i] : 0.0;
            S.x[j][i] = cold ? S.x0r[i] : xs;
            S.nu[j][i] = (first || cold || dead) ? 0.0 : ns;
        }
        {
            const double us = (load && !term) ? a.U[(inst * N + kc) * NU] : 0.0;
            S.u[j] = (term || cold) ? 0.0 : us;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const double l0 = load ? bnd[0 * nb + i] : 0.0, l1 = load ? bnd[1 * nb + i] : 0.0, t0 = load ? bnd[2 * nb + i] : 1.0, t1 = load ? bnd[3 * nb + i] : 1.0;
            S.lam[j][0][i] = cold ? 0.0 : l0, S.lam[j][1][i] = cold ? 0.0 : l1, S.t[j][0][i] = cold ? 1.0 : t0, S.t[j][1][i] = cold ? 1.0 : t1;
            S.aff[j][0][i] = S.aff[j][1][i] = 0.0;
        }
        {
            const int i = NU;
            const double s0 = load ? bnd[4 * nb + i] : 0.0, s1 = load ? bnd[5 * nb + i] : 0.0, m0 = load ? bnd[6 * nb + i] : 0.0, m1 = load ? bnd[7 * nb + i] : 0.0,
                         w0 = load ? bnd[8 * nb + i] : 1.0, w1 = load ? bnd[9 * nb + i] : 1.0;
            S.s[j][0] = cold ? 0.0 : s0, S.s[j][1] = cold ? 0.0 : s1, S.lams[j][0] = cold ? 0.0 : m0, S.lams[j][1] = cold ? 0.0 : m1;
            S.ts[j][0] = cold ? 1.0 : w0, S.ts[j][1] = cold ? 1.0 : w1, S.affs[j][0] = S.affs[j][1] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) S.P[j][i] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) S.p[j][i] = 0.0, S.dx[j][i] = 0.0, S.nuq[j][i] = 0.0, S.Dx[j][i] = 0.0, S.Dnu[j][i] = 0.0, S.K[j][i] = 0.0, S.rb[j][i] = 0.0;
        S.du[j] = 0.0, S.Du[j] = 0.0, S.kff[j] = 0.0, S.Li[j] = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) S.rg[j][i] = 0.0, S.rt[j][i] = 0.0, S.Dg[j][i] = 0.0;
    }
    // ---- full-step SQP (an LQ problem: the first QP is the answer; the loop is the one of small_solve_kernel)
    const bool rti = (a.flags & 4) != 0;
    const int max_iter = rti ? 1 : sp.max_iter;
    bool live = valid, last_tight = true;
    int status = 2, n_sqp = 0, n_ipm = 0;
    double rbest = 1e300, rchk = 1e300;
    int exit_cnt = sp.exit_window;
    double stepn = -1.0;
    if (!(a.flags & (8 | 16))) {
        double sl = 0.0;
        if (S.first[0]) {
            sl = fmax(fabs(S.x0r[0] - S.x[0][0]), fabs(S.x0r[1] - S.x[0][1]));
            if (S.qmode) sl = fmax(sl, fabs(S.u0r - S.u[0]));
        }
        stepn = S.red_max(sl);
        if (cold) stepn = -1.0;
    }
    double Vout = 0.0, res_out[4] = {0, 0, 0, 0};
    for (int it = 0;; ++it) {
        double xin[NX], nuin[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xin[i] = lane_dn(S.x[0][i]), nuin[i] = lane_dn(S.nu[0][i]);
        double res[4] = {0, 0, 0, 0}, cost = 0.0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double *xn = j + 1 < SPL ? S.x[j + 1 < SPL ? j + 1 : 0] : xin, *nun = j + 1 < SPL ? S.nu[j + 1 < SPL ? j + 1 : 0] : nuin;
            cost += S.linearize(j, xn);
            S.nlp_res_local(j, nun, res);
        }
        S.template red<4, 1>(res, &cost);
#ifdef MPCRL_PROFILE_PHASES
        { unsigned long long n_ = clock64(); if (S.pht) S.phw[9] += n_ - S.pht; S.pht = n_; }
#endif
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        if (live) {
            Vout = cost, n_sqp = it;
#pragma unroll
            for (int j = 0; j < 4; ++j) res_out[j] = res[j];
            if (!(rmax < 1e300) || !(fabs(cost) < 1e300))
                status = 1, live = false;
            else if (rmax < sp.tol && last_tight && !(rti && it == 0))
                status = 0, live = false;
            else if (it >= max_iter)
                status = rmax < sp.tol ? 0 : 2, live = false;
            else if (sp.exit_window > 0) {
                rbest = fmin(rbest, rmax);
                if (it == 0)
                    rchk = rmax;
                else if (--exit_cnt == 0) {
                    if (rbest > sp.exit_factor * rchk) status = 2, live = false;
                    rchk = rbest, exit_cnt = sp.exit_window;
                }
            }
        }
        // (LQ model: the QP tolerances are the tight ones from the first QP on)
        const double tol_res = IPM_TOL_RES, tol_mu = IPM_TOL_MU;
        if (live) last_tight = true;
        if (!__any(live)) break;
        // (MPCRL_EXACT_QP: the LQ model's QP is tight and its fraction to the boundary fixed anyway; the flag takes the interior-point warm start away)
        const double warm_mu = (stepn < 0.0 || (a.flags & 64)) ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
        // A warm interior point can jam against rows that the moved x0 makes active, or settle into a two-cycle of mu (Mehrotra's sigma
        // alternating), and run out of iterations; the QP of the LQ model is convex, so the cold start solves it: an instance whose WARM
        // QP failed runs it once more from the cold interior point (same primal iterate; its iterations are counted on top).
        bool ok = false, act = live;
        double wm = warm_mu;
#pragma nounroll
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool o = S.qp_solve(act, n_ipm, wm, tol_res, tol_mu);
            if (act) ok = o;
            act = act && !o && wm > 0.0;
            if (!__any(act)) break;
            wm = 0.0;
        }
        if (live && !ok) status = 4, live = false;
        {
            double sl = 0.0;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                if (S.dead[j]) continue;
                sl = fmax(sl, fmax(fabs(S.dx[j][0]), fabs(S.dx[j][1])));
                if (!S.term[j]) sl = fmax(sl, fabs(S.du[j]));
            }
            stepn = S.red_max(sl);
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
#pragma unroll
                for (int i = 0; i < NX; ++i) S.x[j][i] += S.dx[j][i], S.nu[j][i] = S.nuq[j][i];
                S.u[j] += S.du[j];
            }
        }
    }
#ifdef MPCRL_PROFILE_PHASES
    S.phw[15] = wall_clock64() - rt0_;      // the wavefront's life on the constant 100 MHz clock (the other buckets count shader cycles)
    if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_ticks[i_], S.phw[i_]);
#endif
    // ---- results
    if (a.flags & (1 | 2)) S.sens_tail(a, inst, valid, status);
    double lag = 0.0;
    {
        double nuin[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) nuin[i] = lane_dn(S.nu[0][i]);
        if (a.LAG) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                if (S.dead[j]) continue;
                const double *nun = j + 1 < SPL ? S.nu[j + 1 < SPL ? j + 1 : 0] : nuin;
                if (!S.term[j]) lag = fma(nun[0], S.r_(j, 0), fma(nun[1], S.r_(j, 1), lag));
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if (S.term[j] && i < NU) continue;
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (S.has(j, sd, i)) {
                            lag = fma(-S.lam[j][sd][i], S.bslack(j, sd, i, S.vc(j, i)), lag);
                            if (S.softc(j, i)) lag = fma(-S.lams[j][sd], S.s[j][sd], lag);
                        }
                }
            }
            lag = S.red_sum(lag);
        }
    }
    if (valid && pos == 0) {
        if (a.LAG) a.LAG[inst] = Vout + lag;
        a.u0_out[inst * NU] = S.u[0];
        a.V[inst] = Vout;
        a.status[inst] = status;
        if (a.iters) a.iters[inst * 2] = n_sqp, a.iters[inst * 2 + 1] = n_ipm;
#pragma unroll
        for (int j = 0; j < 4; ++j) a.RES[inst * 4 + j] = res_out[j];
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            if (S.dead[j]) continue;
            const int k = S.kst[j];
            double *bnd = a.BND + (size_t)inst * 10 * nb + (size_t)k * NW;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                a.X[(inst * (N + 1) + k) * NX + i] = S.x[j][i];
                if (!S.first[j]) a.PI[(inst * N + k - 1) * NX + i] = S.nu[j][i];
            }
            if (!S.term[j]) a.U[(inst * N + k) * NU] = S.u[j];
            if (a.flags & 32) continue;      // MPCRL_NO_BND_STORE (wave-uniform): the caller will not warm-start the interior point from this solve
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                bnd[0 * nb + i] = S.has(j, 0, i) ? S.lam[j][0][i] : 0.0, bnd[1 * nb + i] = S.has(j, 1, i) ? S.lam[j][1][i] : 0.0;
                bnd[2 * nb + i] = S.has(j, 0, i) ? S.t[j][0][i] : 1.0, bnd[3 * nb + i] = S.has(j, 1, i) ? S.t[j][1][i] : 1.0;
                const bool sc = i == NU;      // (the coordinate the model can soften; the others keep the values of a cold iterate)
                bnd[4 * nb + i] = sc ? S.s[j][0] : 0.0, bnd[5 * nb + i] = sc ? S.s[j][1] : 0.0;
                bnd[6 * nb + i] = sc ? S.lams[j][0] : 0.0, bnd[7 * nb + i] = sc ? S.lams[j][1] : 0.0;
                bnd[8 * nb + i] = sc ? S.ts[j][0] : 1.0, bnd[9 * nb + i] = sc ? S.ts[j][1] : 1.0;
            }
        }
    }
}

}  // namespace mpcrl
