"""Iterate arrays (the layouts of include/mpcrl.h: x, u, pi, bnd) -> the dense oracle's ``Solution``, so that the mirror of the
reference's NLP (oracle/nlp_mirror.py) can be evaluated AT AN ITERATE IT DID NOT PRODUCE — the product's, pulled through
``mpcrl_get_iterate``, or the C++ port's.

TEST INFRASTRUCTURE (see oracle/__init__.py).

This is what ``update_nlp`` does with the acados solver's output (rlmpc/mpc/nlp.py:1354-1398: copy x, u, pi, lam, t, sl, su out of
the solver stage by stage into the mirror's ``vars`` / ``pi`` / ``lam``), followed by its consistency thresholds
(nlp.py:1445-1537) and the dense solve for dz/dp (nlp.py:1410-1424).  The multiplier of the pinned initial state (the reference
carries it as the pair lbx_0 / ubx_0, nlp.py:648-662) is recovered from the stationarity row of x_0, as acados itself reports it.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .problems import Problem
from .sqp_dense import Solution, Structure


def solution_from_iterate(P: Problem, x, u, pi, bnd, p=None, u0fix=None, gamma=None, cost: Optional[float] = None,
                          res=None, status: int = 0) -> Solution:
    """x (N+1, nx), u (N, nu), pi (N, nx), bnd (10, N+1, nu+nx) of ONE instance."""
    N, nx, nu = P.N, P.nx, P.nu
    x, u, pi, bnd = (np.asarray(a, float) for a in (x, u, pi, bnd))
    p = P.p0 if p is None else np.asarray(p, float)
    st = Structure(P, q_mode=u0fix is not None)
    lam, t = np.zeros(st.nh), np.zeros(st.nh)
    s = np.zeros(st.nS)
    soft_coord = [int(P.idxbx[j]) for j in P.idxsbx]
    for r, (k, kind, j, vi, sgn, b, sv) in enumerate(st.rows):
        if kind in ("lbu", "ubu"):
            side, col = (0 if kind == "lbu" else 1), j
            lam[r], t[r] = bnd[side, k, col], bnd[2 + side, k, col]
        elif kind in ("lbx", "ubx"):
            side = 0 if kind == "lbx" else 1
            col = nu + int((P.idxbx_e if k == N else P.idxbx)[j])
            lam[r], t[r] = bnd[side, k, col], bnd[2 + side, k, col]
        else:   # lsbx / usbx: the rows -s <= 0 of the L1 slacks
            side, col = (0 if kind == "lsbx" else 1), nu + soft_coord[j]
            lam[r], t[r] = bnd[6 + side, k, col], bnd[8 + side, k, col]
            s[(st.isl(k, j) if side == 0 else st.isu(k, j)) - st.nw] = bnd[4 + side, k, col]
    # multiplier of x_0 = x0: stationarity of the Lagrangian in x_0 (no bound rows on stage-0 states)
    c = P.cost_scaling(gamma)
    x0t = torch.tensor(x[0], requires_grad=True)
    pt = torch.tensor(p)
    val = float(c[0]) * P.stage_cost(0, x0t, torch.tensor(u[0]), pt) + torch.tensor(pi[0]) @ P.F(x0t, torch.tensor(u[0]), pt)
    pi0 = -torch.autograd.grad(val, x0t)[0].numpy()
    if cost is None:
        cost = float(c[N]) * float(P.terminal_cost(torch.tensor(x[N]), pt))
        for k in range(N):
            cost += float(c[k]) * float(P.stage_cost(k, torch.tensor(x[k]), torch.tensor(u[k]), pt))
        sw = P.slack_scaling(gamma)
        for k in range(1, N):
            for n in range(st.ns_stage):
                cost += sw[k] * (P.zl[n] * s[st.isl(k, n) - st.nw] + P.zu[n] * s[st.isu(k, n) - st.nw])
    return Solution(status=status, sqp_iter=0, ipm_iter=0, x=x, u=u, pi=pi, pi0=pi0, lam=lam, t=t, s=s, cost=float(cost),
                    res=np.zeros(4) if res is None else np.asarray(res, float), struct=st)


def certify(P: Problem, x, u, pi, bnd, x0, p=None, u0fix=None, gamma=None, cost: Optional[float] = None):
    """Evaluates the mirror at the given iterate, applies the reference's update_nlp thresholds (nlp.py:1445-1537) and returns
    the MirrorResult (dL_dp, dpi_dp = dz_dp[:nu] by the dense solve of nlp.py:1410-1424) with the strict-complementarity margin."""
    from . import nlp_mirror as M
    sol = solution_from_iterate(P, x, u, pi, bnd, p=p, u0fix=u0fix, gamma=gamma, cost=cost)
    mr = M.evaluate(P, sol, x0, p=p, u0fix=u0fix, gamma=gamma)
    M.assert_reference_consistency(P, sol, mr)
    sc = float(np.minimum(np.maximum(sol.lam, sol.t), 1e30).min()) if len(sol.lam) else 1e30
    return mr, sc, sol


def certify_job(job):
    """Process-pool worker of the GPU certification tests (a fresh interpreter): (problem name, kwargs, iterate arrays, x0, p, u0fix,
    gamma, V) -> (dL/dp, dz/dp[:nu], L, strict-complementarity margin, largest slack, stationarity residual); raises if one of the
    reference's update_nlp thresholds fails at the iterate."""
    name, kw, x, u, pi, bnd, x0, p, u0fix, gamma, V = job
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from .problems import make_cartpole, make_chain_mass, make_linear_system
    P = {"cartpole": make_cartpole, "linear": make_linear_system, "chain": make_chain_mass}[name](**kw)
    mr, sc, sol = certify(P, x, u, pi, bnd, x0, p=p, u0fix=u0fix, gamma=gamma, cost=V)
    smax = float(np.abs(sol.s).max()) if len(sol.s) else 0.0
    return mr.dL_dp[0], mr.dpi_dp, mr.L, sc, smax, float(np.abs(mr.R[: P.N * P.nu + (P.N + 1) * P.nx]).max())
