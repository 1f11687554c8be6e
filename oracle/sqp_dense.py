"""O2 — dense full-step SQP with a dense primal-dual interior-point QP solver (numpy + torch autograd).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates what ``ocp_solver.solve()`` does for the reference (call sites
rlmpc/mpc/common/mpc.py:42,79,195; options config/cartpole.yaml:8-14,
linear_system/acados.py:121-124, chain_mass/ocp_utils.py:292-312): SQP with
*full steps* (no globalisation is requested anywhere in the reference), the
Hessian of the stage cost (GAUSS_NEWTON for a (non)linear-least-squares cost that is linear in
(x,u) == the exact cost Hessian; EXACT for the LTI system whose dynamics are
linear == the same thing), a QP per iteration solved by an interior-point
method, termination on the four residuals (stationarity, equality,
inequality, complementarity) < tol.  acados/HPIPM themselves are not vendored
in the reference; their published algorithm (Mehrotra predictor-corrector on
the OCP-structured QP) is restated here with *dense* linear algebra so that
this file shares no factorisation code with the engine (which uses a Riccati
recursion).

The iteration (initial point, step rule, centring rule, stopping rule) is
specified exactly once, here, and is reproduced operation-for-operation by
``oracle/cpu`` (C++) and by the HIP kernels, so the three agree to rounding.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
from torch.func import grad, hessian, jacfwd, vmap

from .problems import Problem

# ---- interior point constants (shared by all three implementations) ----
IPM_MAX_ITER = 60
IPM_TOL_RES = 1e-9      # inf-norm of stationarity / equality / inequality residuals
IPM_TOL_MU = 1e-11      # average complementarity
IPM_T_MIN = 1e-1        # lower clip of the initial slack
IPM_MU0 = 1.0           # lambda_0 = mu0 / t_0
IPM_FRAC = 0.995        # fraction to the boundary: tau = max(IPM_FRAC, 1 - mu) (nearly full steps once the barrier parameter is small)
# warm start of QP j >= 1 (and of the first QP of a warm call) from the previous QP's rows and multipliers:
# every complementarity product is raised to at least mu_w = clamp(IPM_WARM_C * step^2, MIN, MAX), step = inf-norm of
# the previous primal step (or of the change of the pinned x0 / u0 for a warm call)
IPM_WARM_C, IPM_WARM_MIN, IPM_WARM_MAX = 1e-4, 1e-10, 3e-2
# inexact SQP: the QP tolerances follow the NLP residual r, tol_res = clamp(C r^2, TOL_RES, CAP), tol_mu = clamp(C r^2 / 100, TOL_MU, CAP / 10);
# convergence is only declared after a QP that was solved to the tight tolerances
IPM_ADAPT_C, IPM_ADAPT_CAP = 1e1, 3e-2
# Where the affine (predictor) step alone takes the complementarity down by more than two orders of magnitude — sigma = (mu_aff / mu)^3
# below this — it IS the Newton step of a QP that is all but solved (the last iteration of a QP, the warm-started QPs near the SQP's
# limit): the corrector solve is skipped and the predictor step taken.  Problems that opt in (Problem.extra["skip_corrector"]: the
# cartpole — a third of its interior-point iterations, none more needed; not the LQ problem, where it never triggers, nor the chain of
# masses, where iterations are lost), tuned mode only.
IPM_SKIP_SIGMA = 3e-6


@dataclass
class Structure:
    """Index bookkeeping for w = [u_0..u_{N-1}; x_0..x_N; s] and the inequality rows
    in the reference's multiplier order (rlmpc/common/utils.py:4-25, nlp.py:801-819)."""
    prob: Problem
    q_mode: bool = False
    rows: list = field(default_factory=list)     # (stage, kind, local idx, var index, sign, bound, slack var or -1)

    def __post_init__(self):
        P = self.prob
        N, nx, nu = P.N, P.nx, P.nu
        self.nU = N * nu
        self.nX = (N + 1) * nx
        self.ns_stage = len(P.idxsbx)
        self.nS = 2 * self.ns_stage * (N - 1) if self.ns_stage else 0
        self.nw = self.nU + self.nX
        self.nv = self.nw + self.nS
        rows = []
        for k in range(N + 1):
            if k == 0:
                if not self.q_mode:
                    for i in range(nu):
                        rows.append((0, "lbu", i, self.iu(0, i), -1.0, P.lbu[i], -1))
                    for i in range(nu):
                        rows.append((0, "ubu", i, self.iu(0, i), +1.0, P.ubu[i], -1))
            elif k < N:
                for i in range(nu):
                    rows.append((k, "lbu", i, self.iu(k, i), -1.0, P.lbu[i], -1))
                for j, ix in enumerate(P.idxbx):
                    sv = self.isl(k, list(P.idxsbx).index(j)) if j in P.idxsbx else -1
                    rows.append((k, "lbx", j, self.ix(k, ix), -1.0, P.lbx[j], sv))
                for i in range(nu):
                    rows.append((k, "ubu", i, self.iu(k, i), +1.0, P.ubu[i], -1))
                for j, ix in enumerate(P.idxbx):
                    sv = self.isu(k, list(P.idxsbx).index(j)) if j in P.idxsbx else -1
                    rows.append((k, "ubx", j, self.ix(k, ix), +1.0, P.ubx[j], sv))
                for j in range(self.ns_stage):
                    rows.append((k, "lsbx", j, self.isl(k, j), -1.0, 0.0, -2))
                for j in range(self.ns_stage):
                    rows.append((k, "usbx", j, self.isu(k, j), -1.0, 0.0, -2))
            else:
                for j, ix in enumerate(P.idxbx_e):
                    rows.append((k, "lbx", j, self.ix(k, ix), -1.0, P.lbx_e[j], -1))
                for j, ix in enumerate(P.idxbx_e):
                    rows.append((k, "ubx", j, self.ix(k, ix), +1.0, P.ubx_e[j], -1))
        self.rows = rows
        self.nh = len(rows)

    def iu(self, k, i=0):
        return k * self.prob.nu + i

    def ix(self, k, i=0):
        return self.nU + k * self.prob.nx + i

    def isl(self, k, j):
        return self.nw + (k - 1) * 2 * self.ns_stage + j

    def isu(self, k, j):
        return self.nw + (k - 1) * 2 * self.ns_stage + self.ns_stage + j

    def ineq_matrices(self, w):
        """C v <= d  in QP variables v = [delta w; s]  (s absolute)."""
        C = np.zeros((self.nh, self.nv))
        d = np.zeros(self.nh)
        for r, (k, kind, j, vi, sgn, bnd, sv) in enumerate(self.rows):
            if sv == -2:                      # -s <= 0
                C[r, vi] = -1.0
                d[r] = 0.0
                continue
            C[r, vi] = sgn
            d[r] = sgn * (bnd - w[vi])        # sgn*(w+dw) <= sgn*bnd
            if sv >= 0:
                C[r, sv] = -1.0               # lb - x - sl <= 0 ; x - ub - su <= 0
        return C, d


@dataclass
class Solution:
    status: int
    sqp_iter: int
    ipm_iter: int
    x: np.ndarray          # (N+1, nx)
    u: np.ndarray          # (N, nu)
    pi: np.ndarray         # (N, nx)   pi[k] multiplies F(x_k,u_k) - x_{k+1}   (nlp.py:827,1180)
    pi0: np.ndarray        # (nx,)     multiplier of x_0 = x0
    lam: np.ndarray        # (nh,)     reference order
    t: np.ndarray          # (nh,)
    s: np.ndarray          # (nS,)
    cost: float
    res: np.ndarray        # (4,) stat, eq, ineq, comp
    struct: Structure = None


class Linearizer:
    def __init__(self, prob: Problem):
        self.prob = prob
        P = prob
        self._jac = vmap(jacfwd(lambda x, u, p: P.F(x, u, p), argnums=(0, 1)), in_dims=(0, 0, None))
        self._F = vmap(lambda x, u, p: P.F(x, u, p), in_dims=(0, 0, None))

    def dynamics(self, X, U, p):
        Xt, Ut, pt = torch.tensor(X[:-1]), torch.tensor(U), torch.tensor(p)
        A, B = self._jac(Xt, Ut, pt)
        Fv = self._F(Xt, Ut, pt)
        return A.numpy(), B.numpy(), Fv.numpy()

    def cost(self, X, U, p, c):
        """returns value, per-stage gradient (N+1, nu+nx) [u first], per-stage Hessian."""
        P = self.prob
        pt = torch.tensor(p)
        nu, nx, N = P.nu, P.nx, P.N
        g = np.zeros((N + 1, nu + nx))
        H = np.zeros((N + 1, nu + nx, nu + nx))
        val = 0.0
        for k in range(N):
            x = torch.tensor(X[k], requires_grad=True)
            u = torch.tensor(U[k], requires_grad=True)

            def f(v, k=k):
                return P.stage_cost(k, v[nu:], v[:nu], pt)
            v = torch.cat([u, x]).detach()
            val += c[k] * float(f(v))
            g[k] = c[k] * grad(f)(v).numpy()
            H[k] = c[k] * hessian(f)(v).numpy()
        v = torch.tensor(X[N])
        fe = lambda xx: P.terminal_cost(xx, pt)
        val += c[N] * float(fe(v))
        g[N, nu:] = c[N] * grad(fe)(v).numpy()
        H[N, nu:, nu:] = c[N] * hessian(fe)(v).numpy()
        return val, g, H


def ipm_dense(H, g, G, b, C, d, v0, warm=None, free=None, tol_res=IPM_TOL_RES, tol_mu=IPM_TOL_MU, lq=False, skip_corrector=False):
    """Mehrotra predictor-corrector on  min 1/2 v'Hv + g'v  s.t. Gv = b, Cv + t = d, t >= 0.
    warm = (mu_w, lam_prev, t_prev, pi_prev) or None.  free: mask of the variables that are not pinned by an equality
    row of their own (x_0, and u_0 in Q-mode); the stationarity rows of pinned variables only define the multiplier of
    that row and are left out of the stopping test (the structured implementations eliminate them).
    Returns v, pi, lam, t, iterations, ok."""
    n, me, mi = H.shape[0], G.shape[0], C.shape[0]
    v = v0.copy()
    pi = np.zeros(me) if warm is None else warm[3].copy()
    if mi == 0:
        K = np.block([[H, G.T], [G, np.zeros((me, me))]])
        sol = np.linalg.solve(K, np.concatenate([-g, b]))
        return sol[:n], sol[n:], np.zeros(0), np.zeros(0), 1, True
    if warm is None:
        t = np.maximum(d - C @ v, IPM_T_MIN)
        lam = IPM_MU0 / t
    else:
        mu_w, lam, t = warm[0], warm[1].copy(), np.maximum(d - C @ v, warm[2])
        low = lam * t < mu_w
        big_l = lam >= t
        t = np.where(low & big_l, mu_w / np.where(lam > 0, lam, 1.0), t)
        lam = np.where(low & ~big_l, mu_w / t, lam)
    ok = False
    it = 0
    for it in range(IPM_MAX_ITER + 1):
        r_g = H @ v + g + G.T @ pi + C.T @ lam
        r_b = G @ v - b
        r_d = C @ v + t - d
        mu = float(lam @ t) / mi
        rinf = max(np.abs(r_g if free is None else r_g[free]).max(), np.abs(r_b).max() if me else 0.0, np.abs(r_d).max())
        if rinf <= tol_res and mu <= tol_mu:
            ok = True
            break
        if it == IPM_MAX_ITER or not np.isfinite(rinf):
            break
        w_ = lam / t
        K = np.block([[H + C.T @ (w_[:, None] * C), G.T], [G, np.zeros((me, me))]])
        lu = np.linalg.inv(K)     # small systems; explicit inverse keeps the two solves trivially consistent

        def solve(r_m):
            rhs_v = -r_g - C.T @ ((-r_m + lam * r_d) / t)
            sol = lu @ np.concatenate([rhs_v, -r_b])
            dv, dpi = sol[:n], sol[n:]
            dt = -r_d - C @ dv
            dlam = (-r_m - lam * dt) / t
            return dv, dpi, dlam, dt

        def max_step(dlam, dt):
            a = 1.0
            neg = dlam < 0
            if neg.any():
                a = min(a, float((-lam[neg] / dlam[neg]).min()))
            neg = dt < 0
            if neg.any():
                a = min(a, float((-t[neg] / dt[neg]).min()))
            return a

        dv, dpi, dlam, dt = solve(lam * t)
        a_aff = max_step(dlam, dt)
        mu_aff = float((lam + a_aff * dlam) @ (t + a_aff * dt)) / mi
        sigma = (mu_aff / mu) ** 3
        if skip_corrector and sigma < IPM_SKIP_SIGMA:
            a = min(1.0, max(IPM_FRAC, 1.0 - mu) * a_aff)
        else:
            dv, dpi, dlam, dt = solve(lam * t + dlam * dt - sigma * mu)
            a = min(1.0, (IPM_FRAC if lq else max(IPM_FRAC, 1.0 - mu)) * max_step(dlam, dt))   # fraction to the boundary -> 1 as mu -> 0 (not for LQ problems)
        v = v + a * dv
        pi = pi + a * dpi
        lam = lam + a * dlam
        t = t + a * dt
    return v, pi, lam, t, it, ok


def nlp_residuals(prob: Problem, st: Structure, X, U, S, pi0, pi, lam, A, B, Fv, gcost, x0, u0fix, slack_w):
    """stationarity / equality / inequality / complementarity, inf-norms (what acados tests against tol)."""
    N, nx, nu = prob.N, prob.nx, prob.nu
    w = np.concatenate([U.reshape(-1), X.reshape(-1)])
    gL = np.zeros(st.nv)
    for k in range(N + 1):
        if k < N:
            gL[st.iu(k): st.iu(k) + nu] += gcost[k, :nu] + B[k].T @ pi[k]
            gL[st.ix(k): st.ix(k) + nx] += gcost[k, nu:] + A[k].T @ pi[k]
        else:
            gL[st.ix(k): st.ix(k) + nx] += gcost[k, nu:]
        if k > 0:
            gL[st.ix(k): st.ix(k) + nx] -= pi[k - 1]
    gL[st.ix(0): st.ix(0) + nx] += pi0
    if st.nS:
        gL[st.nw:] += slack_w
    v = np.concatenate([w, S])
    C, d = st.ineq_matrices(np.zeros(st.nw))          # absolute form: C v <= d with w = 0 offset
    gL += C.T @ lam
    if u0fix is not None:
        gL[: nu] = 0.0                                # u_0 is not a variable in Q-mode
    hval = C @ v - d
    r_eq = max(np.abs(Fv - X[1:]).max(), np.abs(X[0] - x0).max())
    if u0fix is not None:
        r_eq = max(r_eq, np.abs(U[0] - u0fix).max())
    r_stat = np.abs(gL).max()
    r_ineq = max(0.0, hval.max()) if len(hval) else 0.0
    r_comp = np.abs(lam * hval).max() if len(hval) else 0.0
    return np.array([r_stat, r_eq, r_ineq, r_comp])


def solve(prob: Problem, x0, p=None, u0fix=None, gamma=None, warm: Optional[Solution] = None,
          max_iter=None, tol=None, verbose=False, exact: bool = False, rti: bool = False) -> Solution:
    """Full-step SQP from the reference's cold start (MPC.reset, mpc.py:204-210: x_k = x0, u_k = 0)
    or from ``warm``.  ``u0fix`` reproduces q_update (mpc.py:52-96: lbu_0 = ubu_0 = u0).

    ``exact=True`` is the FROZEN exact-QP mode — what acados + HPIPM do with the reference's options (config/cartpole.yaml:8-14):
    every QP to the tight tolerances (1e-9, 1e-11) from a cold interior-point start, fixed fraction to the boundary 0.995; no
    forcing term, no warm start, no adaptive step rule.  The tuned (inexact) iteration is tested against it
    (tests/test_exact_vs_inexact.py); its constants must not change when the product is tuned.
    ``rti=True``: exactly one SQP iteration from ``warm`` (the product's MPCRL_RTI; not a reference mode)."""
    if rti:
        max_iter = 1
    P = prob
    N, nx, nu = P.N, P.nx, P.nu
    p = P.p0 if p is None else np.asarray(p, float)
    x0 = np.asarray(x0, float).reshape(nx)
    tol = P.tol if tol is None else tol
    max_iter = P.max_iter if max_iter is None else max_iter
    st = Structure(P, q_mode=u0fix is not None)
    lin = Linearizer(P)
    c = P.cost_scaling(gamma)
    sw = P.slack_scaling(gamma)
    slack_w = np.zeros(st.nS)
    for k in range(1, N):
        for j in range(st.ns_stage):
            slack_w[st.isl(k, j) - st.nw] = sw[k] * P.zl[j]
            slack_w[st.isu(k, j) - st.nw] = sw[k] * P.zu[j]
    if warm is None:
        X = np.tile(x0, (N + 1, 1))
        U = np.zeros((N, nu))
        S = np.zeros(st.nS)
        pi = np.zeros((N, nx))
        pi0 = np.zeros(nx)
        lam = np.zeros(st.nh)
        t = np.ones(st.nh)
    else:
        X, U, S, pi, pi0 = warm.x.copy(), warm.u.copy(), warm.s.copy(), warm.pi.copy(), warm.pi0.copy()
        lam = warm.lam.copy() if len(warm.lam) == st.nh else np.zeros(st.nh)
        t = warm.t.copy() if len(warm.t) == st.nh else np.ones(st.nh)
    if u0fix is not None:
        u0fix = np.asarray(u0fix, float).reshape(nu)
    # equality rows: x_0 block, then N dynamics blocks, (then u_0 block in Q-mode)
    me = nx + N * nx + (nu if u0fix is not None else 0)
    status, ipm_total = 2, 0
    res = np.full(4, np.inf)
    it = 0
    last_tight = True
    stepn = -1.0                      # < 0: no previous QP to start from
    piq = None
    if warm is not None:
        stepn = float(np.abs(x0 - X[0]).max())
        if u0fix is not None:
            stepn = max(stepn, float(np.abs(np.asarray(u0fix, float).reshape(nu) - U[0]).max()))
        piq = np.concatenate([pi0, pi.reshape(-1)] + ([np.zeros(nu)] if u0fix is not None else []))
    for it in range(max_iter + 1):
        A, B, Fv = lin.dynamics(X, U, p)
        cost, gcost, Hc = lin.cost(X, U, p, c)
        cost += float(slack_w @ S)
        res = nlp_residuals(P, st, X, U, S, pi0, pi, lam, A, B, Fv, gcost, x0, u0fix, slack_w)
        if verbose:
            print(f"sqp {it:3d} cost {cost:.10e} res {res}")
        if not (np.all(np.isfinite(res)) and np.isfinite(cost)):
            status = 1
            break
        if res.max() < tol and last_tight and not (rti and it == 0):
            status = 0
            break
        if it == max_iter:
            status = 0 if res.max() < tol else 2
            break
        rr = min(1.0, float(res.max()))
        a_ = 0.0 if (res.max() < tol or P.extra.get("lq", False) or exact) else IPM_ADAPT_C * rr * rr   # an LQ problem is solved by its first QP
        tol_res = min(IPM_ADAPT_CAP, max(IPM_TOL_RES, a_))
        tol_mu = min(float(P.extra.get("tol_mu_factor", 0.1)) * IPM_ADAPT_CAP, max(IPM_TOL_MU, 1e-2 * a_))   # (chain of masses: factor 1)
        last_tight = tol_res <= IPM_TOL_RES and tol_mu <= IPM_TOL_MU
        # ---- QP in v = [du; dx; s]
        H = np.zeros((st.nv, st.nv))
        g = np.zeros(st.nv)
        G = np.zeros((me, st.nv))
        b = np.zeros(me)
        G[:nx, st.ix(0): st.ix(0) + nx] = np.eye(nx)
        b[:nx] = x0 - X[0]
        for k in range(N + 1):
            if k < N:
                iu, ix = st.iu(k), st.ix(k)
                H[iu: iu + nu, iu: iu + nu] = Hc[k, :nu, :nu]
                H[iu: iu + nu, ix: ix + nx] = Hc[k, :nu, nu:]
                H[ix: ix + nx, iu: iu + nu] = Hc[k, nu:, :nu]
                H[ix: ix + nx, ix: ix + nx] = Hc[k, nu:, nu:]
                g[iu: iu + nu] = gcost[k, :nu]
                g[ix: ix + nx] = gcost[k, nu:]
                r0 = nx + k * nx
                G[r0: r0 + nx, ix: ix + nx] = A[k]
                G[r0: r0 + nx, iu: iu + nu] = B[k]
                G[r0: r0 + nx, st.ix(k + 1): st.ix(k + 1) + nx] = -np.eye(nx)
                b[r0: r0 + nx] = -(Fv[k] - X[k + 1])
            else:
                ix = st.ix(k)
                H[ix: ix + nx, ix: ix + nx] = Hc[k, nu:, nu:]
                g[ix: ix + nx] = gcost[k, nu:]
        if st.nS:
            g[st.nw:] = slack_w
        if u0fix is not None:
            r0 = nx + N * nx
            G[r0: r0 + nu, : nu] = np.eye(nu)
            b[r0: r0 + nu] = u0fix - U[0]
        w = np.concatenate([U.reshape(-1), X.reshape(-1)])
        C, d = st.ineq_matrices(w)
        v0 = np.zeros(st.nv)
        v0[st.ix(0): st.ix(0) + nx] = x0 - X[0]
        if u0fix is not None:
            v0[:nu] = u0fix - U[0]
        wrm = None
        if stepn >= 0.0 and not exact:
            v0[st.nw:] = S
            wrm = (min(IPM_WARM_MAX, max(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn)), lam, t, piq)
        free = np.ones(st.nv, bool)
        free[st.ix(0): st.ix(0) + nx] = False
        if u0fix is not None:
            free[:nu] = False
        v, piq, lam, t, nit, ok = ipm_dense(H, g, G, b, C, d, v0, wrm, free, tol_res, tol_mu, lq=bool(P.extra.get("lq", False)) or exact,
                                              skip_corrector=bool(P.extra.get("skip_corrector", False)) and not exact)
        stepn = float(np.abs(v[: st.nw]).max())
        ipm_total += nit
        if not ok:
            status = 4
            break
        U = U + v[: st.nU].reshape(N, nu)
        X = X + v[st.nU: st.nw].reshape(N + 1, nx)
        S = v[st.nw:].copy()
        pi0 = piq[:nx].copy()
        pi = piq[nx: nx + N * nx].reshape(N, nx).copy()
    return Solution(status=status, sqp_iter=it, ipm_iter=ipm_total, x=X, u=U, pi=pi, pi0=pi0, lam=lam, t=t, s=S,
                    cost=cost, res=res, struct=st)
