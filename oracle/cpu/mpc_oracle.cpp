// CPU oracle port of the MPC-as-policy hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
//
// Serial, structure-exploiting restatement of the algorithm specified in oracle/sqp_dense.py (the dense
// Python oracle is the specification; this file reproduces its iteration with a Riccati recursion instead
// of dense KKT solves) plus the exact-Hessian adjoint sweep for dV/dp and du0*/dp whose definition is
// rlmpc/mpc/nlp.py:1211,1214-1224,1399-1424.  It is what bench.py times as the "port" CPU baseline and what
// the GPU parity tests compare against at sizes the Python oracle cannot reach.  It shares no source with
// mpc4rl_amd/csrc.
#include "models.hpp"

using namespace oracle_impl;

namespace {

enum { LB = 0, UB = 1, SL = 2, SU = 3 };

template <class Mdl>
struct Solver {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NW = NX + NU, NTD = Mdl::NTD, NP = Mdl::NP;
    const OracleSpec &sp;
    const int N, n;
    const double *p;        // full parameter vector of this instance (rebound per instance by bind())
    double th[NTD];         // dynamics parameters
    std::vector<double> c;  // cost scaling c_k
    std::vector<double> X, U, PI;                       // iterate; PI[k] multiplies F(x_k,u_k) - x_{k+1}
    std::vector<double> A, B, r, q, H;                  // linearisation
    std::vector<double> lb, ub, zw[2], s[2], rgs[2], ds[2];   // bounds, soft weights, slacks (per side)
    std::vector<char> has[2], soft;
    std::vector<double> lam[4], t[4], rd[4], dlam[4], dt[4], aff[4];   // inequality rows by kind
    std::vector<double> dx, du, pi_qp, Dx, Du, Dpi, rg, rb, rt, Dg;
    std::vector<double> P, pv, K, kff, Lc;              // Riccati
    bool qmode = false;
    int n_rows = 0;

    void bind(const double *p_) {
        p = p_;
        for (int i = 0; i < NTD; ++i) th[i] = p[Mdl::td_index(i)];
    }
    Solver(const OracleSpec &sp_, const double *p_) : sp(sp_), N(sp_.N), n((sp_.N + 1) * NW), p(p_) {
        for (int i = 0; i < NTD; ++i) th[i] = p[Mdl::td_index(i)];
        c.resize(N + 1);
        if (sp.cost_kind == ORACLE_COST_NLS) {
            for (int k = 0; k < N; ++k) c[k] = sp.dT;
            c[N] = 1.0;
        } else {
            c[0] = sp.dT;
            for (int k = 1; k < N; ++k) c[k] = std::pow(sp.gamma, k) * sp.dT;
            c[N] = std::pow(sp.gamma, N);
        }
        X.assign((N + 1) * NX, 0), U.assign(N * NU, 0), PI.assign(N * NX, 0);
        A.resize(N * NX * NX), B.resize(N * NX * NU), r.resize(N * NX), q.resize(n), H.resize(n * NW);
        lb.assign(n, -1e30), ub.assign(n, 1e30), soft.assign(n, 0);
        for (int sd = 0; sd < 2; ++sd)
            zw[sd].assign(n, 0), s[sd].assign(n, 0), rgs[sd].assign(n, 0), ds[sd].assign(n, 0), has[sd].assign(n, 0);
        for (int j = 0; j < 4; ++j)
            lam[j].assign(n, 0), t[j].assign(n, 1), rd[j].assign(n, 0), dlam[j].assign(n, 0), dt[j].assign(n, 0), aff[j].assign(n, 0);
        dx.resize((N + 1) * NX), du.resize(N * NU), pi_qp.resize(N * NX), Dx.resize((N + 1) * NX), Du.resize(N * NU);
        Dpi.resize(N * NX), rb.resize(N * NX), rg.assign(n, 0), rt.assign(n, 0), Dg.assign(n, 0);
        P.resize((N + 1) * NX * NX), pv.resize((N + 1) * NX), K.resize(N * NU * NX), kff.resize(N * NU), Lc.resize(N * NU * NU);
    }

    bool active(int kind, int e) const { return has[kind & 1][e] && (kind < 2 || soft[e]); }
    bool skip(int e) const { return e / NW == N && e % NW < NU; }   // no control at stage N

    void setup_bounds(bool qm) {
        qmode = qm;
        n_rows = 0;
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NW; ++i) {
                const int e = k * NW + i;
                double l = -1e30, u = 1e30;
                int sf = 0;
                if (k == 0) {
                    if (i < NU && !qm) l = sp.lb0[i], u = sp.ub0[i];
                } else if (k < N) {
                    l = sp.lb[i], u = sp.ub[i];
                    sf = sp.soft ? sp.soft[i] : 0;
                } else if (i >= NU) {
                    l = sp.lbe[i - NU], u = sp.ube[i - NU];
                }
                lb[e] = l, ub[e] = u;
                has[0][e] = l > -INF_BOUND, has[1][e] = u < INF_BOUND;
                soft[e] = sf;
                if (sf) {   // weight dT * gamma^k * z  (nlp.py:1118-1130)
                    const double w = sp.dT * std::pow(sp.gamma, k);
                    zw[0][e] = w * sp.zl[i], zw[1][e] = w * sp.zu[i];
                }
                for (int j = 0; j < 4; ++j) n_rows += active(j, e);
            }
    }

    double *xk(int k) { return &X[k * NX]; }
    double vcoord(int k, int i) const { return i < NU ? U[k * NU + i] : X[k * NX + i - NU]; }
    double dvc(const std::vector<double> &ax, const std::vector<double> &au, int k, int i) const {
        return i < NU ? (k < N ? au[k * NU + i] : 0.0) : ax[k * NX + i - NU];
    }
    bool is_fixed(int k, int i) const { return k == 0 && (i >= NU || qmode); }

    // ---- linearisation at (X,U): A, B, r = F - x+, q = c grad l, H = c hess l; returns the cost
    double linearize() {
        typedef Dual<double, NW> D1;
        double cost = 0;
        for (int k = 0; k < N; ++k) {
            D1 x[NX], u[NU], tt[NTD], xn[NX];
            for (int i = 0; i < NU; ++i) u[i] = D1(U[k * NU + i]), u[i].d[i] = 1.0;
            for (int i = 0; i < NX; ++i) x[i] = D1(X[k * NX + i]), x[i].d[NU + i] = 1.0;
            for (int i = 0; i < NTD; ++i) tt[i] = D1(th[i]);
            disc_map<Mdl, D1>(x, u, tt, xn, sp);
            for (int i = 0; i < NX; ++i) {
                r[k * NX + i] = xn[i].v - X[(k + 1) * NX + i];
                for (int j = 0; j < NU; ++j) B[(k * NX + i) * NU + j] = xn[i].d[j];
                for (int j = 0; j < NX; ++j) A[(k * NX + i) * NX + j] = xn[i].d[NU + j];
            }
        }
        for (int k = 0; k <= N; ++k) {
            const double *u = k < N ? &U[k * NU] : nullptr;
            cost += c[k] * Mdl::cost_val(k, N, xk(k), u, p, sp);
            Mdl::cost_grad(k, N, xk(k), u, p, sp, &q[k * NW]);
            Mdl::cost_hess(k, N, xk(k), u, p, sp, &H[k * NW * NW]);
            for (int i = 0; i < NW; ++i) q[k * NW + i] *= c[k];
            for (int i = 0; i < NW * NW; ++i) H[k * NW * NW + i] *= c[k];
        }
        for (int e = 0; e < n; ++e)
            if (soft[e]) cost += zw[0][e] * s[0][e] + zw[1][e] * s[1][e];
        return cost;
    }

    // (G' pi) on stage k, coordinate i
    double GTpi(const std::vector<double> &pi, int k, int i) const {
        double a = 0;
        if (k < N) {
            if (i < NU)
                for (int m = 0; m < NX; ++m) a += B[(k * NX + m) * NU + i] * pi[k * NX + m];
            else
                for (int m = 0; m < NX; ++m) a += A[(k * NX + m) * NX + i - NU] * pi[k * NX + m];
        }
        if (i >= NU && k > 0) a -= pi[(k - 1) * NX + i - NU];
        return a;
    }

    // slack(v) of row `kind` on coordinate e at value v of the coordinate
    double row_slack(int kind, int e, double v) const {
        switch (kind) {
            case LB: return v + (soft[e] ? s[0][e] : 0.0) - lb[e];
            case UB: return ub[e] - v + (soft[e] ? s[1][e] : 0.0);
            case SL: return s[0][e];
            default: return s[1][e];
        }
    }

    void nlp_residuals(const double *x0, const double *u0fix, double *res) {
        double rs = 0, re = 0, ri = 0, rc = 0;
        for (int e = 0; e < n; ++e) {
            if (skip(e)) continue;
            const int k = e / NW, i = e % NW;
            if (!is_fixed(k, i)) {
                double g = q[e] + GTpi(PI, k, i);
                if (has[0][e]) g -= lam[LB][e];
                if (has[1][e]) g += lam[UB][e];
                rs = std::max(rs, std::fabs(g));
            }
            for (int j = 0; j < 4; ++j)
                if (active(j, e)) {
                    const double h = -row_slack(j, e, vcoord(k, i));
                    ri = std::max(ri, h), rc = std::max(rc, std::fabs(lam[j][e] * h));
                }
            for (int sd = 0; sd < 2; ++sd)
                if (active(SL + sd, e)) rs = std::max(rs, std::fabs(zw[sd][e] - lam[LB + sd][e] - lam[SL + sd][e]));
        }
        for (int i = 0; i < N * NX; ++i) re = std::max(re, std::fabs(r[i]));
        for (int i = 0; i < NX; ++i) re = std::max(re, std::fabs(X[i] - x0[i]));
        if (u0fix)
            for (int i = 0; i < NU; ++i) re = std::max(re, std::fabs(U[i] - u0fix[i]));
        res[0] = rs, res[1] = re, res[2] = ri, res[3] = rc;
    }

    // ---- Riccati factorisation with stage Hessians Hs + diag(Dd); false if a pivot is not positive
    bool riccati_factor(const double *Hs, const double *Dd) {
        double BA[NX * NW], T[NX * NW], Mm[NW * NW];
        for (int i = 0; i < NX; ++i)
            for (int j = 0; j < NX; ++j)
                P[(N * NX + i) * NX + j] = Hs[(N * NW + NU + i) * NW + NU + j] + (i == j ? Dd[N * NW + NU + i] : 0.0);
        for (int k = N - 1; k >= 0; --k) {
            const double *Pn = &P[(k + 1) * NX * NX];
            for (int m = 0; m < NX; ++m) {
                for (int j = 0; j < NU; ++j) BA[m * NW + j] = B[(k * NX + m) * NU + j];
                for (int j = 0; j < NX; ++j) BA[m * NW + NU + j] = A[(k * NX + m) * NX + j];
            }
            for (int i = 0; i < NX; ++i)
                for (int j = 0; j < NW; ++j) {
                    double a = 0;
                    for (int m = 0; m < NX; ++m) a += Pn[i * NX + m] * BA[m * NW + j];
                    T[i * NW + j] = a;
                }
            for (int i = 0; i < NW; ++i)
                for (int j = 0; j < NW; ++j) {
                    double a = Hs[(k * NW + i) * NW + j] + (i == j ? Dd[k * NW + i] : 0.0);
                    for (int m = 0; m < NX; ++m) a += BA[m * NW + i] * T[m * NW + j];
                    Mm[i * NW + j] = a;
                }
            double *Kk = &K[k * NU * NX], *Lk = &Lc[k * NU * NU];
            if (k == 0 && qmode) {
                for (int i = 0; i < NU * NX; ++i) Kk[i] = 0;
                for (int i = 0; i < NU * NU; ++i) Lk[i] = 0;
            } else {
                for (int i = 0; i < NU; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double a = Mm[i * NW + j];
                        for (int m = 0; m < j; ++m) a -= Lk[i * NU + m] * Lk[j * NU + m];
                        if (i == j) {
                            if (!(a > 0)) return false;
                            Lk[i * NU + i] = std::sqrt(a);
                        } else
                            Lk[i * NU + j] = a / Lk[j * NU + j];
                    }
                for (int j = 0; j < NX; ++j) {   // K = Rt^{-1} S
                    double y[NU];
                    for (int i = 0; i < NU; ++i) {
                        double a = Mm[i * NW + NU + j];
                        for (int m = 0; m < i; ++m) a -= Lk[i * NU + m] * y[m];
                        y[i] = a / Lk[i * NU + i];
                    }
                    for (int i = NU - 1; i >= 0; --i) {
                        double a = y[i];
                        for (int m = i + 1; m < NU; ++m) a -= Lk[m * NU + i] * Kk[m * NX + j];
                        Kk[i * NX + j] = a / Lk[i * NU + i];
                    }
                }
            }
            double *Pk = &P[k * NX * NX];
            for (int i = 0; i < NX; ++i)
                for (int j = 0; j < NX; ++j) {
                    double a = Mm[(NU + i) * NW + NU + j];
                    for (int m = 0; m < NU; ++m) a -= Mm[m * NW + NU + i] * Kk[m * NX + j];
                    Pk[i * NX + j] = a;
                }
            for (int i = 0; i < NX; ++i)
                for (int j = 0; j < i; ++j) {
                    const double sy = 0.5 * (Pk[i * NX + j] + Pk[j * NX + i]);
                    Pk[i * NX + j] = Pk[j * NX + i] = sy;
                }
        }
        return true;
    }

    // ---- min 1/2 v'(H+D)v + g'v  s.t.  x+ = A x + B u + bb,  x_0 step = 0 (u_0 step = 0 in Q-mode)
    void riccati_solve(const double *g, const double *bb, double *ox, double *ou, double *opi) {
        for (int i = 0; i < NX; ++i) pv[N * NX + i] = g[N * NW + NU + i];
        for (int k = N - 1; k >= 0; --k) {
            const double *Pn = &P[(k + 1) * NX * NX];
            double cc[NX], m[NW];
            for (int i = 0; i < NX; ++i) {
                double a = pv[(k + 1) * NX + i];
                for (int j = 0; j < NX; ++j) a += Pn[i * NX + j] * bb[k * NX + j];
                cc[i] = a;
            }
            for (int i = 0; i < NW; ++i) {
                double a = g[k * NW + i];
                for (int mm = 0; mm < NX; ++mm) a += (i < NU ? B[(k * NX + mm) * NU + i] : A[(k * NX + mm) * NX + i - NU]) * cc[mm];
                m[i] = a;
            }
            const double *Lk = &Lc[k * NU * NU], *Kk = &K[k * NU * NX];
            double *kf = &kff[k * NU];
            if (k == 0 && qmode) {
                for (int i = 0; i < NU; ++i) kf[i] = 0;
            } else {
                double y[NU];
                for (int i = 0; i < NU; ++i) {
                    double a = m[i];
                    for (int mm = 0; mm < i; ++mm) a -= Lk[i * NU + mm] * y[mm];
                    y[i] = a / Lk[i * NU + i];
                }
                for (int i = NU - 1; i >= 0; --i) {
                    double a = y[i];
                    for (int mm = i + 1; mm < NU; ++mm) a -= Lk[mm * NU + i] * kf[mm];
                    kf[i] = a / Lk[i * NU + i];
                }
            }
            for (int i = 0; i < NX; ++i) {
                double a = m[NU + i];
                for (int mm = 0; mm < NU; ++mm) a -= Kk[mm * NX + i] * m[mm];
                pv[k * NX + i] = a;
            }
        }
        for (int i = 0; i < NX; ++i) ox[i] = 0;
        for (int k = 0; k < N; ++k) {
            const double *Kk = &K[k * NU * NX];
            for (int i = 0; i < NU; ++i) {
                double a = -kff[k * NU + i];
                for (int j = 0; j < NX; ++j) a -= Kk[i * NX + j] * ox[k * NX + j];
                ou[k * NU + i] = a;
            }
            for (int i = 0; i < NX; ++i) {
                double a = bb[k * NX + i];
                for (int j = 0; j < NX; ++j) a += A[(k * NX + i) * NX + j] * ox[k * NX + j];
                for (int j = 0; j < NU; ++j) a += B[(k * NX + i) * NU + j] * ou[k * NU + j];
                ox[(k + 1) * NX + i] = a;
            }
            const double *Pn = &P[(k + 1) * NX * NX];
            for (int i = 0; i < NX; ++i) {
                double a = pv[(k + 1) * NX + i];
                for (int j = 0; j < NX; ++j) a += Pn[i * NX + j] * ox[(k + 1) * NX + j];
                opi[k * NX + i] = a;
            }
        }
    }

    void eq_residual() {
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < NX; ++i) {
                double a = r[k * NX + i] - dx[(k + 1) * NX + i];
                for (int j = 0; j < NX; ++j) a += A[(k * NX + i) * NX + j] * dx[k * NX + j];
                for (int j = 0; j < NU; ++j) a += B[(k * NX + i) * NU + j] * du[k * NU + j];
                rb[k * NX + i] = a;
            }
    }

    // ---- Mehrotra predictor-corrector (the iteration of oracle/sqp_dense.py:ipm_dense); returns iterations
    // warm_mu > 0: start from the rows (lam, t, s) and multipliers of the previous QP, re-centred so that every
    // complementarity product is at least warm_mu (the smaller factor of the pair is raised); warm_mu = 0: cold start.
    double qp_tol_res = IPM_TOL_RES, qp_tol_mu = IPM_TOL_MU;
    bool exact = false;   // ORACLE_EXACT: tight QPs, cold interior-point starts, fixed fraction to the boundary
    int qp_solve(const double *x0, const double *u0fix, bool &ok, double warm_mu) {
        std::fill(dx.begin(), dx.end(), 0.0), std::fill(du.begin(), du.end(), 0.0), std::fill(pi_qp.begin(), pi_qp.end(), 0.0);
        if (warm_mu > 0.0) pi_qp = PI;
        for (int i = 0; i < NX; ++i) dx[i] = x0[i] - X[i];
        if (u0fix)
            for (int i = 0; i < NU; ++i) du[i] = u0fix[i] - U[i];
        for (int e = 0; e < n; ++e) {
            if (skip(e)) continue;
            const double v = vcoord(e / NW, e % NW) + dvc(dx, du, e / NW, e % NW);
            if (warm_mu > 0.0) {
                for (int j = 0; j < 4; ++j)
                    if (active(j, e)) {
                        double l = lam[j][e], tt = std::max(row_slack(j, e, v), t[j][e]);
                        if (l * tt < warm_mu) {
                            if (l >= tt)
                                tt = warm_mu / l;
                            else
                                l = warm_mu / tt;
                        }
                        lam[j][e] = l, t[j][e] = tt;
                    }
                continue;
            }
            s[0][e] = s[1][e] = 0;
            for (int j = 0; j < 4; ++j)
                if (active(j, e)) t[j][e] = std::max(row_slack(j, e, v), IPM_T_MIN), lam[j][e] = IPM_MU0 / t[j][e];
        }
        ok = false;
        if (n_rows == 0) {   // equality-constrained LQ problem: a single Riccati solve
            std::fill(Dg.begin(), Dg.end(), 0.0);
            eq_residual();
            for (int e = 0; e < n; ++e) {
                double a = q[e];
                for (int j = 0; j < NW; ++j) a += H[e * NW + j] * dvc(dx, du, e / NW, j);
                rt[e] = is_fixed(e / NW, e % NW) ? 0.0 : a;
            }
            if (!riccati_factor(H.data(), Dg.data())) return 1;
            riccati_solve(rt.data(), rb.data(), Dx.data(), Du.data(), Dpi.data());
            for (size_t i = 0; i < dx.size(); ++i) dx[i] += Dx[i];
            for (size_t i = 0; i < du.size(); ++i) du[i] += Du[i];
            pi_qp = Dpi;
            ok = true;
            return 1;
        }
        int it = 0, snap_it = -1;
        double snap_rinf = 1e300;
        std::vector<double> snap_dx, snap_du, snap_pi, snap_lam[4], snap_t[4], snap_s[2];
        for (it = 0; it <= IPM_MAX_ITER; ++it) {
            double rinf = 0, mu = 0;
            eq_residual();
            for (int i = 0; i < N * NX; ++i) rinf = std::max(rinf, std::fabs(rb[i]));
            for (int e = 0; e < n; ++e) {
                rg[e] = 0;
                if (skip(e)) continue;
                const int k = e / NW, i = e % NW;
                double a = q[e] + GTpi(pi_qp, k, i);
                for (int j = 0; j < NW; ++j) a += H[e * NW + j] * dvc(dx, du, k, j);
                if (has[0][e]) a -= lam[LB][e];
                if (has[1][e]) a += lam[UB][e];
                if (is_fixed(k, i)) a = 0;
                rg[e] = a;
                rinf = std::max(rinf, std::fabs(a));
                const double v = vcoord(k, i) + dvc(dx, du, k, i);
                for (int j = 0; j < 4; ++j)
                    if (active(j, e)) {
                        rd[j][e] = t[j][e] - row_slack(j, e, v);
                        rinf = std::max(rinf, std::fabs(rd[j][e]));
                        mu += lam[j][e] * t[j][e];
                    }
                for (int sd = 0; sd < 2; ++sd)
                    if (active(SL + sd, e)) {
                        rgs[sd][e] = zw[sd][e] - lam[LB + sd][e] - lam[SL + sd][e];
                        rinf = std::max(rinf, std::fabs(rgs[sd][e]));
                    }
            }
            mu /= n_rows;
            if (rinf <= qp_tol_res && mu <= qp_tol_mu) {
                ok = true;
                break;
            }
            // Noise floor (round 6).  The residuals are re-evaluated from the iterate here, so they stop at its rounding level — 2e-9 ... 7e-9
            // on warm-started linear-system QPs with an active L1-soft row (penalty 100) — while the kernels carry them forward by the
            // (1 - alpha) scaling, which has no floor: with IPM_TOL_RES = 1e-9 this loop then kept halving mu to 1e-29 and broke down
            // (status 4 on ~1 % of such calls, status 0 from both kernel families: profiles/r06_fuzz_parity.txt).  The best iterate that
            // is complementary to tolerance and within 10 x the residual tolerance is remembered; if the loop ends WITHOUT converging it is
            // what the QP returns.  Nothing changes for a QP that converges.
            if (mu <= qp_tol_mu && rinf <= 10.0 * qp_tol_res && rinf < snap_rinf) {
                snap_rinf = rinf, snap_it = it;
                snap_dx = dx, snap_du = du, snap_pi = pi_qp;
                for (int j = 0; j < 4; ++j) snap_lam[j] = lam[j], snap_t[j] = t[j];
                for (int sd = 0; sd < 2; ++sd) snap_s[sd] = s[sd];
            }
            if (it == IPM_MAX_ITER || !std::isfinite(rinf)) break;
            for (int e = 0; e < n; ++e) {
                double d = 0;
                for (int sd = 0; sd < 2; ++sd)
                    if (has[sd][e]) {
                        const double w1 = lam[LB + sd][e] / t[LB + sd][e];
                        if (soft[e]) {
                            const double w2 = lam[SL + sd][e] / t[SL + sd][e];
                            d += w1 * w2 / (w1 + w2);
                        } else
                            d += w1;
                    }
                Dg[e] = d;
            }
            if (!riccati_factor(H.data(), Dg.data())) break;
            double alpha = 1.0, sigma = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                const double smu = pass ? sigma * mu : 0.0;
                auto rm = [&](int j, int e) { return lam[j][e] * t[j][e] + (pass ? aff[j][e] : 0.0) - smu; };
                auto ee = [&](int j, int e) { return (rm(j, e) - lam[j][e] * rd[j][e]) / t[j][e]; };
                for (int e = 0; e < n; ++e) {
                    double a = rg[e];
                    for (int sd = 0; sd < 2; ++sd)
                        if (has[sd][e]) {
                            const double sg = sd ? -1.0 : 1.0;
                            const double e1 = ee(LB + sd, e);
                            if (soft[e]) {
                                const double e2 = ee(SL + sd, e);
                                const double w1 = lam[LB + sd][e] / t[LB + sd][e], w2 = lam[SL + sd][e] / t[SL + sd][e];
                                a += sg * (e1 * w2 - w1 * (rgs[sd][e] + e2)) / (w1 + w2);
                            } else
                                a += sg * e1;
                        }
                    rt[e] = a;
                }
                riccati_solve(rt.data(), rb.data(), Dx.data(), Du.data(), Dpi.data());
                double amax = 1.0;
                for (int e = 0; e < n; ++e) {
                    if (skip(e)) continue;
                    const double dv = dvc(Dx, Du, e / NW, e % NW);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has[sd][e]) {
                            const double sg = sd ? -1.0 : 1.0;
                            double dss = 0;
                            if (soft[e]) {
                                const double e1 = ee(LB + sd, e), e2 = ee(SL + sd, e);
                                const double w1 = lam[LB + sd][e] / t[LB + sd][e], w2 = lam[SL + sd][e] / t[SL + sd][e];
                                dss = -(rgs[sd][e] + e1 + e2 + sg * w1 * dv) / (w1 + w2);
                                ds[sd][e] = dss;
                                dt[SL + sd][e] = -rd[SL + sd][e] + dss;
                            }
                            dt[LB + sd][e] = -rd[LB + sd][e] + sg * dv + dss;
                        }
                    for (int j = 0; j < 4; ++j)
                        if (active(j, e)) {
                            dlam[j][e] = (-rm(j, e) - lam[j][e] * dt[j][e]) / t[j][e];
                            if (dlam[j][e] < 0) amax = std::min(amax, -lam[j][e] / dlam[j][e]);
                            if (dt[j][e] < 0) amax = std::min(amax, -t[j][e] / dt[j][e]);
                        }
                }
                if (pass == 0) {
                    double sum = 0;
                    for (int e = 0; e < n; ++e)
                        for (int j = 0; j < 4; ++j)
                            if (!skip(e) && active(j, e)) {
                                sum += (lam[j][e] + amax * dlam[j][e]) * (t[j][e] + amax * dt[j][e]);
                                aff[j][e] = dlam[j][e] * dt[j][e];
                            }
                    const double ratio = (sum / n_rows) / mu;
                    sigma = ratio * ratio * ratio;
                    if (Mdl::SKIP_CORRECTOR && !exact && sigma < IPM_SKIP_SIGMA) {   // the predictor step is the step (sqp_dense.py IPM_SKIP_SIGMA)
                        alpha = std::min(1.0, std::max(IPM_FRAC, 1.0 - mu) * amax);
                        break;
                    }
                } else
                    alpha = std::min(1.0, ((Mdl::DISCRETE || exact) ? IPM_FRAC : std::max(IPM_FRAC, 1.0 - mu)) * amax);   // fraction to the boundary -> 1 as mu -> 0 (LQ model: fixed)
            }
            for (size_t i = 0; i < dx.size(); ++i) dx[i] += alpha * Dx[i];
            for (size_t i = 0; i < du.size(); ++i) du[i] += alpha * Du[i];
            for (size_t i = 0; i < pi_qp.size(); ++i) pi_qp[i] += alpha * Dpi[i];
            for (int e = 0; e < n; ++e) {
                if (skip(e)) continue;
                for (int j = 0; j < 4; ++j)
                    if (active(j, e)) lam[j][e] += alpha * dlam[j][e], t[j][e] += alpha * dt[j][e];
                for (int sd = 0; sd < 2; ++sd)
                    if (active(SL + sd, e)) s[sd][e] += alpha * ds[sd][e];
            }
        }
        if (!ok && snap_it >= 0) {      // (see "noise floor" above)
            dx = snap_dx, du = snap_du, pi_qp = snap_pi;
            for (int j = 0; j < 4; ++j) lam[j] = snap_lam[j], t[j] = snap_t[j];
            for (int sd = 0; sd < 2; ++sd) s[sd] = snap_s[sd];
            ok = true;
            return snap_it;
        }
        return it;
    }

    // ---- full-step SQP (oracle/sqp_dense.py:solve)
    // stall > 0: the product's opt-in divergence exit (mpcrl_set_exit_rule): every `stall` SQP iterations the best NLP residual
    // seen so far must have dropped below `stall_factor` times its value at the previous check, else the instance ends with
    // status 2 (not a reference behaviour: the reference runs full-step SQP to max_iter, config/cartpole.yaml:12-14)
    int sqp(const double *x0, const double *u0fix, bool warm, int max_iter, double tol, double *res, int &n_sqp, int &n_ipm,
            double &cost, bool rti = false, int stall = 0, double stall_factor = 0.1) {
        if (rti) max_iter = 1;
        setup_bounds(u0fix != nullptr);
        if (!warm) {
            for (int k = 0; k <= N; ++k)
                for (int i = 0; i < NX; ++i) X[k * NX + i] = x0[i];   // MPC.reset, mpc.py:204-210
            std::fill(U.begin(), U.end(), 0.0), std::fill(PI.begin(), PI.end(), 0.0);
            for (int j = 0; j < 4; ++j) std::fill(lam[j].begin(), lam[j].end(), 0.0), std::fill(t[j].begin(), t[j].end(), 1.0);
            for (int sd = 0; sd < 2; ++sd) std::fill(s[sd].begin(), s[sd].end(), 0.0);
        }
        n_ipm = 0;
        int status = 2;
        // size of the perturbation the next QP sees: change of the pinned initial state (warm call), then the last step
        bool last_tight = true;
        double rbest = 1e300, rchk = 1e300;
        double stepn = -1.0;   // < 0: no previous QP to start from
        if (warm) {
            stepn = 0.0;
            for (int i = 0; i < NX; ++i) stepn = std::max(stepn, std::fabs(x0[i] - X[i]));
            if (u0fix)
                for (int i = 0; i < NU; ++i) stepn = std::max(stepn, std::fabs(u0fix[i] - U[i]));
        }
        for (n_sqp = 0;; ++n_sqp) {
            cost = linearize();
            nlp_residuals(x0, u0fix, res);
            const double rmax = std::max(std::max(res[0], res[1]), std::max(res[2], res[3]));
            if (!std::isfinite(rmax) || !std::isfinite(cost)) return 1;   // std::max drops NaNs, the cost sum does not
            if (rmax < tol && last_tight && !(rti && n_sqp == 0)) return 0;
            if (n_sqp == max_iter) return rmax < tol ? 0 : 2;
            rbest = std::min(rbest, rmax);
            if (stall > 0 && n_sqp > 0 && n_sqp % stall == 0) {
                if (rbest > stall_factor * rchk) return 2;
                rchk = rbest;
            }
            if (n_sqp == 0) rchk = rmax;
            {   // QP tolerances for this iteration
                // (a linear-quadratic OCP is solved by its first QP: no inexactness there)
                const double rr = std::min(1.0, rmax), a = (rmax < tol || Mdl::DISCRETE || exact) ? 0.0 : IPM_ADAPT_C * rr * rr;
                qp_tol_res = std::min(IPM_ADAPT_CAP, std::max(IPM_TOL_RES, a));
                qp_tol_mu = std::min(Mdl::TOL_MU_FACTOR * IPM_ADAPT_CAP, std::max(IPM_TOL_MU, 1e-2 * a));
                last_tight = qp_tol_res <= IPM_TOL_RES && qp_tol_mu <= IPM_TOL_MU;
            }
            bool ok;
            const double warm_mu = (stepn < 0.0 || exact) ? 0.0 : std::min(IPM_WARM_MAX, std::max(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
            n_ipm += qp_solve(x0, u0fix, ok, warm_mu);
            // (round 6, LQ model: a WARM interior point that ran out of iterations — jammed against rows the moved x0 activates
            // (alpha 1e-13 ... 1e-6 for 15 iterations from mu 2e-8 under a residual of 2e-2), then a two-cycle of mu with sigma
            // alternating 0.07 / 0.88 — once more from the cold interior point, which solves the convex QP; the kernels do the same)
            if (!ok && warm_mu > 0.0 && Mdl::DISCRETE) n_ipm += qp_solve(x0, u0fix, ok, 0.0);
            if (!ok) return 4;
            stepn = 0.0;
            for (double v : dx) stepn = std::max(stepn, std::fabs(v));
            for (double v : du) stepn = std::max(stepn, std::fabs(v));
            for (size_t i = 0; i < X.size(); ++i) X[i] += dx[i];
            for (size_t i = 0; i < U.size(); ++i) U[i] += du[i];
            PI = pi_qp;
        }
        return status;
    }

    // ---- dV/dp = dL/dp (nlp.py:1211,1401) and du0*/dp (nlp.py:1413-1424) by an adjoint Riccati solve
    void sensitivities(int flags, double *dV, double *dpi) {
        std::vector<double> Fth(N * NX * NTD);
        {
            typedef Dual<double, NTD> D1;
            for (int k = 0; k < N; ++k) {
                D1 x[NX], u[NU], tt[NTD], xn[NX];
                for (int i = 0; i < NU; ++i) u[i] = D1(U[k * NU + i]);
                for (int i = 0; i < NX; ++i) x[i] = D1(X[k * NX + i]);
                for (int i = 0; i < NTD; ++i) tt[i] = D1(th[i]), tt[i].d[i] = 1.0;
                disc_map<Mdl, D1>(x, u, tt, xn, sp);
                for (int m = 0; m < NX; ++m)
                    for (int d = 0; d < NTD; ++d) Fth[(k * NX + m) * NTD + d] = xn[m].d[d];
            }
        }
        if (dV && (flags & ORACLE_SENS_V)) {
            for (int i = 0; i < NP; ++i) dV[i] = 0;
            for (int k = 0; k < N; ++k)
                for (int d = 0; d < NTD; ++d) {
                    double a = 0;
                    for (int m = 0; m < NX; ++m) a += PI[k * NX + m] * Fth[(k * NX + m) * NTD + d];
                    dV[Mdl::td_index(d)] += a;
                }
            for (int k = 0; k <= N; ++k) Mdl::cost_dp(k, N, xk(k), k < N ? &U[k * NU] : nullptr, p, sp, c[k], dV);
        }
        if (!(dpi && (flags & ORACLE_SENS_PI))) return;
        for (int i = 0; i < NU * NP; ++i) dpi[i] = 0;
        if (qmode) return;   // u_0 is pinned: lbu_0 = ubu_0 = u0 (mpc.py:71-76)
        // exact Lagrangian Hessian blocks (nlp.py:1202,1224): c hess l + sum_m pi_m hess F_m
        std::vector<double> Hex(H);
        {
            typedef Dual<double, 1> In;
            typedef Dual<In, NW> D2;
            for (int k = 0; k < N; ++k)
                for (int j = 0; j < NW; ++j) {
                    D2 x[NX], u[NU], tt[NTD], xn[NX];
                    for (int i = 0; i < NU; ++i) u[i] = D2(U[k * NU + i]), u[i].d[i].v = 1.0;
                    for (int i = 0; i < NX; ++i) x[i] = D2(X[k * NX + i]), x[i].d[NU + i].v = 1.0;
                    for (int i = 0; i < NTD; ++i) tt[i] = D2(th[i]);
                    if (j < NU)
                        u[j].v.d[0] = 1.0;
                    else
                        x[j - NU].v.d[0] = 1.0;
                    disc_map<Mdl, D2>(x, u, tt, xn, sp);
                    for (int i = 0; i < NW; ++i) {
                        double a = 0;
                        for (int m = 0; m < NX; ++m) a += PI[k * NX + m] * xn[m].d[i].d[0];
                        Hex[(k * NW + i) * NW + j] += a;
                    }
                }
        }
        // barrier diagonal from the final (lam, t) of the bound rows; slack variables are constants (quirk q1)
        for (int e = 0; e < n; ++e) {
            double d = 0;
            for (int sd = 0; sd < 2; ++sd)
                if (has[sd][e]) d += std::min(lam[LB + sd][e] / t[LB + sd][e], SENS_W_MAX);
            Dg[e] = d;
        }
        if (!riccati_factor(Hex.data(), Dg.data())) {
            for (int i = 0; i < NU * NP; ++i) dpi[i] = NAN;
            return;
        }
        std::vector<double> zero(N * NX, 0.0), yx((N + 1) * NX), yu(N * NU), ypi(N * NX), yv(NW);
        // Where the cap binds on a STATE row (an active state bound: lam / t ~ 1e20) the capped solve carries a bias c / W in the adjoint
        // solution — 1.5e-6 ... 3.6e-6 of du0/dp on cartpole states with the cart at the end of its track (G7b) — which is linear in
        // 1 / W down to W ~ 1e11, while the recursion's rounding grows with W.  Richardson in the cap removes it at W's rounding level:
        // y = 2 y(W) - y(W / 2).  Per instance, only for the models whose kernels do the same (Mdl::SENS_EXTRAP), only where a state row is capped.
        static_assert(!Mdl::SENS_EXTRAP || NU == 1, "the second factorisation below replaces the first: one adjoint solve only");
        bool extrap = false;
        if (Mdl::SENS_EXTRAP)
            for (int e = 0; e < n; ++e)
                if (e % NW >= NU)
                    for (int sd = 0; sd < 2; ++sd)
                        if (has[sd][e] && lam[LB + sd][e] / t[LB + sd][e] > SENS_W_MAX) extrap = true;
        std::vector<double> yx2, yu2, ypi2;
        for (int iu = 0; iu < NU; ++iu) {
            std::fill(rt.begin(), rt.end(), 0.0);
            rt[iu] = -1.0;
            riccati_solve(rt.data(), zero.data(), yx.data(), yu.data(), ypi.data());
            if (extrap) {
                yx2.resize(yx.size()), yu2.resize(yu.size()), ypi2.resize(ypi.size());
                for (int e = 0; e < n; ++e) {
                    double d = 0;
                    for (int sd = 0; sd < 2; ++sd)
                        if (has[sd][e]) d += std::min(lam[LB + sd][e] / t[LB + sd][e], 0.5 * SENS_W_MAX);
                    Dg[e] = d;
                }
                if (!riccati_factor(Hex.data(), Dg.data())) {
                    for (int i = 0; i < NU * NP; ++i) dpi[i] = NAN;
                    return;
                }
                riccati_solve(rt.data(), zero.data(), yx2.data(), yu2.data(), ypi2.data());
                for (size_t i = 0; i < yx.size(); ++i) yx[i] = 2.0 * yx[i] - yx2[i];
                for (size_t i = 0; i < yu.size(); ++i) yu[i] = 2.0 * yu[i] - yu2[i];
                for (size_t i = 0; i < ypi.size(); ++i) ypi[i] = 2.0 * ypi[i] - ypi2[i];
            }
            double *out = dpi + iu * NP;
            typedef Dual<double, 1> In;
            typedef Dual<In, NTD> D2;
            for (int k = 0; k <= N; ++k) {
                for (int i = 0; i < NW; ++i) yv[i] = dvc(yx, yu, k, i);
                std::vector<double> tmp(NP, 0.0);
                Mdl::cost_mixed(k, N, xk(k), k < N ? &U[k * NU] : nullptr, p, sp, yv.data(), c[k], tmp.data());
                for (int i = 0; i < NP; ++i) out[i] -= tmp[i];
                if (k == N) break;
                D2 x[NX], u[NU], tt[NTD], xn[NX];
                for (int i = 0; i < NU; ++i) u[i] = D2(U[k * NU + i]), u[i].v.d[0] = yv[i];
                for (int i = 0; i < NX; ++i) x[i] = D2(X[k * NX + i]), x[i].v.d[0] = yv[NU + i];
                for (int i = 0; i < NTD; ++i) tt[i] = D2(th[i]), tt[i].d[i].v = 1.0;
                disc_map<Mdl, D2>(x, u, tt, xn, sp);
                for (int d = 0; d < NTD; ++d) {
                    double a = 0;
                    for (int m = 0; m < NX; ++m)
                        a += PI[k * NX + m] * xn[m].d[d].d[0] + ypi[k * NX + m] * Fth[(k * NX + m) * NTD + d];
                    out[Mdl::td_index(d)] -= a;
                }
            }
        }
    }
};

template <class Mdl>
int run(const OracleSpec *sp, int Bn, const double *x0, const double *u0fix, const double *p, int ppi, int flags, double *X,
        double *U, double *PI, double *BND, double *u0_out, double *V, double *dV, double *dpi, int *status, int *sqp_iter,
        int *ipm_iter, double *res, int nthreads) {
    if (sp->nx != Mdl::NX || sp->nu != Mdl::NU || sp->np != Mdl::NP) return -2;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NW = NX + NU, NP = Mdl::NP;
    const int N = sp->N, n = (N + 1) * NW;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
    Solver<Mdl> S(*sp, p);   // one workspace per thread, re-used for every instance the thread picks up
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int b = 0; b < Bn; ++b) {
        S.bind(ppi ? p + (size_t)b * NP : p);
        const bool warm = flags & ORACLE_WARM;
        if (warm) {
            std::copy(X + (size_t)b * (N + 1) * NX, X + (size_t)(b + 1) * (N + 1) * NX, S.X.begin());
            std::copy(U + (size_t)b * N * NU, U + (size_t)(b + 1) * N * NU, S.U.begin());
            std::copy(PI + (size_t)b * N * NX, PI + (size_t)(b + 1) * N * NX, S.PI.begin());
            if (BND) {
                const double *bd = BND + (size_t)b * 10 * n;
                std::copy(bd, bd + n, S.lam[LB].begin()), std::copy(bd + n, bd + 2 * n, S.lam[UB].begin());
                std::copy(bd + 2 * n, bd + 3 * n, S.t[LB].begin()), std::copy(bd + 3 * n, bd + 4 * n, S.t[UB].begin());
                std::copy(bd + 4 * n, bd + 5 * n, S.s[0].begin()), std::copy(bd + 5 * n, bd + 6 * n, S.s[1].begin());
                std::copy(bd + 6 * n, bd + 7 * n, S.lam[SL].begin()), std::copy(bd + 7 * n, bd + 8 * n, S.lam[SU].begin());
                std::copy(bd + 8 * n, bd + 9 * n, S.t[SL].begin()), std::copy(bd + 9 * n, bd + 10 * n, S.t[SU].begin());
            }
        }
        double r4[4], cost = 0;
        int ns = 0, ni = 0;
        S.exact = (flags & ORACLE_EXACT) != 0;
        const int st = S.sqp(x0 + (size_t)b * NX, u0fix ? u0fix + (size_t)b * NU : nullptr, warm, sp->max_iter, sp->tol, r4, ns, ni, cost,
                             (flags & ORACLE_RTI) != 0, sp->exit_window, sp->exit_factor);
        if (status) status[b] = st;
        if (sqp_iter) sqp_iter[b] = ns;
        if (ipm_iter) ipm_iter[b] = ni;
        if (res) std::copy(r4, r4 + 4, res + (size_t)b * 4);
        if (V) V[b] = cost;
        if (u0_out) std::copy(S.U.begin(), S.U.begin() + NU, u0_out + (size_t)b * NU);
        if (X) std::copy(S.X.begin(), S.X.end(), X + (size_t)b * (N + 1) * NX);
        if (U) std::copy(S.U.begin(), S.U.end(), U + (size_t)b * N * NU);
        if (PI) std::copy(S.PI.begin(), S.PI.end(), PI + (size_t)b * N * NX);
        if (BND) {
            double *bd = BND + (size_t)b * 10 * n;
            std::copy(S.lam[LB].begin(), S.lam[LB].end(), bd), std::copy(S.lam[UB].begin(), S.lam[UB].end(), bd + n);
            std::copy(S.t[LB].begin(), S.t[LB].end(), bd + 2 * n), std::copy(S.t[UB].begin(), S.t[UB].end(), bd + 3 * n);
            std::copy(S.s[0].begin(), S.s[0].end(), bd + 4 * n), std::copy(S.s[1].begin(), S.s[1].end(), bd + 5 * n);
            std::copy(S.lam[SL].begin(), S.lam[SL].end(), bd + 6 * n), std::copy(S.lam[SU].begin(), S.lam[SU].end(), bd + 7 * n);
            std::copy(S.t[SL].begin(), S.t[SL].end(), bd + 8 * n), std::copy(S.t[SU].begin(), S.t[SU].end(), bd + 9 * n);
        }
        if ((flags & (ORACLE_SENS_V | ORACLE_SENS_PI)) && (st == 0 || st == 2))
            S.sensitivities(flags, dV ? dV + (size_t)b * NP : nullptr, dpi ? dpi + (size_t)b * NU * NP : nullptr);
    }
    }
    return 0;
}

}  // namespace

extern "C" int mpc_oracle_solve(const OracleSpec *sp, int B, const double *x0, const double *u0fix, const double *p,
                                int p_per_instance, int flags, double *X, double *U, double *PI, double *BND, double *u0_out,
                                double *V, double *dV, double *dpi, int *status, int *sqp_iter, int *ipm_iter, double *res,
                                int nthreads) {
    if (!sp || !x0 || !p || B < 0) return -1;
    if ((flags & ORACLE_WARM) && !(X && U && PI)) return -1;
#define ARGS sp, B, x0, u0fix, p, p_per_instance, flags, X, U, PI, BND, u0_out, V, dV, dpi, status, sqp_iter, ipm_iter, res, nthreads
    switch (sp->model) {
        case ORACLE_MODEL_CARTPOLE: return run<Cartpole>(ARGS);
        case ORACLE_MODEL_LINEAR: return run<Linear>(ARGS);
        case ORACLE_MODEL_CHAIN:
            if (sp->nx == Chain<5>::NX) return run<Chain<5>>(ARGS);
            if (sp->nx == Chain<7>::NX) return run<Chain<7>>(ARGS);
            if (sp->nx == Chain<3>::NX) return run<Chain<3>>(ARGS);
            if (sp->nx == Chain<4>::NX) return run<Chain<4>>(ARGS);
            if (sp->nx == Chain<6>::NX) return run<Chain<6>>(ARGS);
            return -2;
    }
#undef ARGS
    return -2;
}
