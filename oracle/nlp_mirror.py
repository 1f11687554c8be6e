"""O1 — restatement of the reference's NLP mirror and of what dV/dp and dpi/dp *are*.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows rlmpc/mpc/nlp.py:
  * vars / p / pi / lam / t layouts ............... nlp.py:903-1010
  * g_k = F(x_k,u_k,p_model) - x_{k+1} ............ nlp.py:822-831
  * h <= 0 rows, per stage, reference order ....... nlp.py:644-819, common/utils.py:4-25
    (stage 0 carries lbu_0, lbx_0, ubu_0, ubx_0 with lbx_0 = ubx_0 = x0, nlp.py:648-662, 1374-1375)
  * L = cost + lam.h + pi.g ....................... nlp.py:1180
  * w = [u_0..u_{N-1}; x_0..x_N],  z = [w; pi; lam; t] ... nlp.py:1190-1201,1220
  * R = [grad_w L; g; h + t; lam*t - tau], tau=1e-8 .. nlp.py:1199,1214
  * dL_dp = dL/dp ................................. nlp.py:1211,1401
  * dz_dp = solve(dR_dz, -dR_dp); dpi_dp = dz_dp[:nu] . nlp.py:1413-1424
The slack variables sl/su are *constants* of the mirror (not in z) — quirk q1.

Derivatives come from torch autograd of R, not from the engine's formulas.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
from torch.func import grad, jacrev

from .problems import Problem
from .sqp_dense import Solution

TAU = 1e-8   # nlp.py:1199


@dataclass
class MirrorResult:
    dL_dp: np.ndarray     # (1, n_p)
    dpi_dp: np.ndarray    # (nu, n_p)
    R: np.ndarray         # KKT residual at the solution
    L: float
    cost: float
    h: np.ndarray
    lam: np.ndarray
    t: np.ndarray
    dR_dz: np.ndarray
    dR_dp: np.ndarray


def mirror_rows(prob: Problem):
    """(stage, var kind, index, sign, slack position or -1/-2) in the reference order; the bound VALUE is
    looked up at evaluation time because lbx_0/ubx_0 (and lbu_0/ubu_0 in Q-mode) are run-time data."""
    P = prob
    rows = []
    for k in range(P.N + 1):
        if k == 0:
            rows += [(0, "u", i, -1.0, "lbu0", -1) for i in range(P.nu)]
            rows += [(0, "x", i, -1.0, "lbx0", -1) for i in range(P.nx)]
            rows += [(0, "u", i, +1.0, "ubu0", -1) for i in range(P.nu)]
            rows += [(0, "x", i, +1.0, "ubx0", -1) for i in range(P.nx)]
        elif k < P.N:
            soft = {int(j): n for n, j in enumerate(P.idxsbx)}
            rows += [(k, "u", i, -1.0, "lbu", -1) for i in range(P.nu)]
            rows += [(k, "x", int(ix), -1.0, ("lbx", j), ("sl", soft[j]) if j in soft else -1) for j, ix in enumerate(P.idxbx)]
            rows += [(k, "u", i, +1.0, "ubu", -1) for i in range(P.nu)]
            rows += [(k, "x", int(ix), +1.0, ("ubx", j), ("su", soft[j]) if j in soft else -1) for j, ix in enumerate(P.idxbx)]
            rows += [(k, "s", ("sl", n), -1.0, None, -2) for n in range(len(P.idxsbx))]
            rows += [(k, "s", ("su", n), -1.0, None, -2) for n in range(len(P.idxsbx))]
        else:
            rows += [(k, "x", int(ix), -1.0, ("lbx_e", j), -1) for j, ix in enumerate(P.idxbx_e)]
            rows += [(k, "x", int(ix), +1.0, ("ubx_e", j), -1) for j, ix in enumerate(P.idxbx_e)]
    return rows


def evaluate(prob: Problem, sol: Solution, x0, p=None, u0fix=None, gamma=None) -> MirrorResult:
    P = prob
    N, nx, nu = P.N, P.nx, P.nu
    p = P.p0 if p is None else np.asarray(p, float)
    x0 = np.asarray(x0, float).reshape(nx)
    c = torch.tensor(P.cost_scaling(gamma))
    sw = P.slack_scaling(gamma)
    rows = mirror_rows(P)
    nh = len(rows)
    ns = len(P.idxsbx)
    st = sol.struct

    # slack values per stage (constants of the mirror)
    def s_val(k, which, n):
        return sol.s[(st.isl(k, n) if which == "sl" else st.isu(k, n)) - st.nw]

    # ---- bound values + selection matrices so that h(w) = Jh w + h0 (all rows are affine in w)
    nw = N * nu + (N + 1) * nx
    Jh = np.zeros((nh, nw))
    h0 = np.zeros(nh)
    lbu0 = P.lbu if u0fix is None else np.asarray(u0fix, float)
    ubu0 = P.ubu if u0fix is None else np.asarray(u0fix, float)
    for r, (k, kind, idx, sgn, bnd, sv) in enumerate(rows):
        if kind == "s":
            h0[r] = -s_val(k, idx[0], idx[1])
            continue
        col = k * nu + idx if kind == "u" else N * nu + k * nx + idx
        Jh[r, col] = sgn
        if bnd == "lbu0":
            b = lbu0[idx]
        elif bnd == "ubu0":
            b = ubu0[idx]
        elif bnd in ("lbx0", "ubx0"):
            b = x0[idx]
        elif bnd == "lbu":
            b = P.lbu[idx]
        elif bnd == "ubu":
            b = P.ubu[idx]
        else:
            b = getattr(P, bnd[0])[bnd[1]]
        h0[r] = -sgn * b
        if sv != -1:
            h0[r] -= s_val(k, sv[0], sv[1])
    Jh_t, h0_t = torch.tensor(Jh), torch.tensor(h0)

    # ---- slack penalty (constant wrt w and p): nlp.py:1118-1130
    slack_cost = 0.0
    for k in range(1, N):
        for n in range(ns):
            slack_cost += sw[k] * (P.zl[n] * s_val(k, "sl", n) + P.zu[n] * s_val(k, "su", n))

    def split(w):
        U = w[: N * nu].reshape(N, nu)
        X = w[N * nu:].reshape(N + 1, nx)
        return U, X

    def cost_fn(w, pp):
        U, X = split(w)
        val = c[N] * P.terminal_cost(X[N], pp)
        for k in range(N):
            val = val + c[k] * P.stage_cost(k, X[k], U[k], pp)
        return val + slack_cost

    def g_fn(w, pp):
        U, X = split(w)
        return torch.cat([P.F(X[k], U[k], pp) - X[k + 1] for k in range(N)])

    def h_fn(w):
        return Jh_t @ w + h0_t

    def L_fn(w, pi, lam, pp):
        return cost_fn(w, pp) + lam @ h_fn(w) + pi @ g_fn(w, pp)

    def R_fn(z, pp):
        w = z[:nw]
        pi = z[nw: nw + N * nx]
        lam = z[nw + N * nx: nw + N * nx + nh]
        t = z[nw + N * nx + nh:]
        dLdw = grad(L_fn, argnums=0)(w, pi, lam, pp)
        return torch.cat([dLdw, g_fn(w, pp), h_fn(w) + t, lam * t - TAU])

    # ---- multipliers in mirror order
    w_np = np.concatenate([sol.u.reshape(-1), sol.x.reshape(-1)])
    lam = np.zeros(nh)
    t = np.zeros(nh)
    # map solver rows (no stage-0 x rows; no stage-0 u rows in Q-mode) into the mirror
    key = {}
    for r, (k, kind, j, vi, sgn, bnd, sv) in enumerate(st.rows):
        key[(k, kind, j)] = r
    # equality-like pairs get lam_ub - lam_lb = nu_eq, t = tau / lam  (see DESIGN.md "x0 as two inequalities")
    if u0fix is not None:
        wt = torch.tensor(w_np, requires_grad=True)
        pit = torch.tensor(sol.pi.reshape(-1))
        gr = torch.autograd.grad(cost_fn(wt, torch.tensor(p)) + pit @ g_fn(wt, torch.tensor(p)), wt)[0].numpy()
        nu_u0 = -gr[:nu]
    cnt_kind = {}
    for r, (k, kind, idx, sgn, bnd, sv) in enumerate(rows):
        if bnd in ("lbx0", "ubx0"):
            v = sol.pi0[idx]
            lam[r] = (max(v, 0.0) if sgn > 0 else max(-v, 0.0)) + 1.0
            t[r] = TAU / lam[r]
        elif bnd in ("lbu0", "ubu0") and u0fix is not None:
            v = nu_u0[idx]
            lam[r] = (max(v, 0.0) if sgn > 0 else max(-v, 0.0)) + 1.0
            t[r] = TAU / lam[r]
        else:
            if kind == "s":
                name = "lsbx" if idx[0] == "sl" else "usbx"
                j = idx[1]
            elif kind == "u":
                name = "lbu" if sgn < 0 else "ubu"
                j = idx
            else:
                name = "lbx" if sgn < 0 else "ubx"
                j = bnd[1]
            rr = key[(k, name, j)]
            lam[r] = sol.lam[rr]
            t[r] = sol.t[rr]
    z = torch.tensor(np.concatenate([w_np, sol.pi.reshape(-1), lam, t]))
    pt = torch.tensor(p)
    w_t = z[:nw]
    pi_t = z[nw: nw + N * nx]
    lam_t = z[nw + N * nx: nw + N * nx + nh]
    R = R_fn(z, pt)
    dL_dp = grad(L_fn, argnums=3)(w_t, pi_t, lam_t, pt)
    dR_dz = jacrev(R_fn, argnums=0)(z, pt).numpy()
    dR_dp = jacrev(R_fn, argnums=1)(z, pt).numpy()
    dz_dp = np.linalg.solve(dR_dz, -dR_dp)
    return MirrorResult(
        dL_dp=dL_dp.numpy().reshape(1, -1), dpi_dp=dz_dp[:nu, :], R=R.numpy(),
        L=float(L_fn(w_t, pi_t, lam_t, pt)), cost=float(cost_fn(w_t, pt)), h=h_fn(w_t).numpy(), lam=lam, t=t,
        dR_dz=dR_dz, dR_dp=dR_dp,
    )


def assert_reference_consistency(prob: Problem, sol: Solution, mr: MirrorResult):
    """The thresholds ``update_nlp`` applies to every solver output (nlp.py:1445-1537)."""
    assert abs(mr.cost - sol.cost) < 1e-3                                  # nlp.py:1445-1448
    nw = prob.N * prob.nu + (prob.N + 1) * prob.nx
    g = mr.R[nw: nw + prob.N * prob.nx]
    assert np.abs(g).max() <= 1e-4                                         # nlp.py:1514-1517
    assert np.all(mr.h < 1e-6)                                             # nlp.py:1523
    assert np.abs(mr.lam * mr.h).max() <= 1e-5                             # nlp.py:1525-1527
    assert np.allclose(mr.R[:nw], 0.0, atol=1e-3)                          # nlp.py:1529-1537
