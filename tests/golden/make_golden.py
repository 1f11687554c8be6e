"""Generates the committed golden fixtures in tests/golden/ (run in the dev container: `python tests/golden/make_golden.py`).

The reference itself cannot run here or on the GPU box (acados / CasADi are not vendored and not installable),
and it stores no expected outputs, so these vectors come from the build's own independent oracle:
  * G1  closed-form discounted LQR / DARE known answers (scipy), SURVEY.md §8c "G1"
  * G2  linear_system  (rlmpc/mpc/linear_system/acados.py, tests/test_linear_example.py params)
  * G3  cartpole N=20  (config/cartpole.yaml, reset distribution of continuous_cartpole/environment.py:178-180)
  * G4  chain_mass n_mass=5: the parameter sweep of tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174
  * G5  chain_mass n_mass=7 (nx=33, the perf dimension of BASELINE config 4): nominal + two perturbed initial states
each solved by oracle/sqp_dense.py in its FROZEN EXACT-QP MODE (dense SQP, every QP solved cold to 1e-9 / 1e-11 — what the
reference's acados + HPIPM settings do, config/cartpole.yaml:8-14) with sensitivities from oracle/nlp_mirror.py (torch autograd
of the residual R(z,p) of nlp.py:1214 + dense solve, nlp.py:1410-1424).  The product's tuned inexact iteration never produced
these numbers.  The NLP tolerance is GOLD_TOL = 1e-8, not the solver default 1e-6: the vectors are the KKT POINT itself, so that a
run stopped at tol = 1e-6 (the reference's setting, and the product's default) can be compared with it — two SQP runs that both
stop at 1e-6 agree with each other only to ~1e-6 x conditioning (measured: up to 4e-6 relative in u0* between the dense oracle and
its own C++ port), which says nothing about either.  The chain problems (2385 / 3400 KKT unknowns, minutes per dense SQP iteration)
start the dense iteration from the C++ port's exact-mode solution, so the dense oracle only has to certify it (its own residual
evaluation, its own QP solves) in one or two iterations.  Inputs and expected outputs only — no reference source.
`python tests/golden/make_golden.py [g1 g2 ...]` (G3 is 385 dense solves: ~3 minutes on 6 processes).
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "1")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import nlp_mirror as M  # noqa: E402
from oracle import sqp_dense as S  # noqa: E402
from oracle.problems import make_cartpole, make_chain_mass, make_linear_system  # noqa: E402


def g1():
    from scipy.linalg import solve_discrete_are
    A = np.array([[1.0, 0.25], [0.0, 1.0]])
    B = np.array([[0.03125], [0.25]])
    P = solve_discrete_are(A, B, np.eye(2), np.eye(1))
    x0 = np.array([0.5, 0.5])
    out = {"A": A.tolist(), "B": B.tolist(), "P_dare": P.tolist(), "x0": x0.tolist(), "V_0": 1e-3, "N": 40, "cases": []}
    for gamma in (1.0, 0.99, 0.9):
        Pk = P.copy() * gamma ** 40           # terminal weight gamma^N P; stage weights gamma^k I (nlp.py:1083-1091)
        K = None
        for k in range(39, -1, -1):
            c = gamma ** k
            Sm = c * np.eye(1) + B.T @ Pk @ B
            K = np.linalg.solve(Sm, B.T @ Pk @ A)
            Pk = c * np.eye(2) + A.T @ Pk @ A - A.T @ Pk @ B @ K
        out["cases"].append({"gamma": gamma, "u0": float((-K @ x0)[0]), "V": float(1e-3 + 0.5 * x0 @ Pk @ x0)})
    json.dump(out, open(os.path.join(HERE, "g1_lqr.json"), "w"), indent=1)
    print("g1", out["cases"])


KEYS = ["u0", "V", "dV", "dpi", "X", "U", "PI", "sqp_iter", "ipm_iter", "res", "status", "smax", "L", "sc"]


GOLD_TOL = 1e-8      # the exact mode solves its QPs to 1e-9 / 1e-11: the NLP complementarity residual bottoms out at ~2e-9


def _one(job):
    name, kw, i, x0, p, u0, gamma = job
    import torch
    torch.set_num_threads(1)
    P = {"cartpole": make_cartpole, "chain": make_chain_mass, "linear": make_linear_system}[name](**kw)
    warm = None
    if name == "chain":
        from oracle import cpu_port
        pr = cpu_port.solve(P, x0[None], p=None if p is None else p[None], u0fix=None if u0 is None else u0[None], gamma=gamma,
                            exact=True, tol=GOLD_TOL, flags=0, nthreads=1)
        assert pr.status[0] == 0
        warm = S.Solution(status=0, sqp_iter=0, ipm_iter=0, x=pr.X[0], u=pr.U[0], pi=pr.PI[0], pi0=np.zeros(P.nx), lam=np.zeros(0),
                          t=np.zeros(0), s=np.zeros(0), cost=float(pr.V[0]), res=pr.res[0])
    sol = S.solve(P, x0, p=p, u0fix=u0, gamma=gamma, exact=True, tol=GOLD_TOL, warm=warm)
    assert sol.status == 0 and np.all(sol.res < GOLD_TOL), (sol.status, sol.res)
    mr = M.evaluate(P, sol, x0, p=p, u0fix=u0, gamma=gamma)
    M.assert_reference_consistency(P, sol, mr)
    print(P.name, i, sol.status, sol.sqp_iter, sol.ipm_iter, sol.u[0], sol.cost, flush=True)
    return [sol.u[0], sol.cost, mr.dL_dp[0], mr.dpi_dp, sol.x, sol.u, sol.pi, sol.sqp_iter, sol.ipm_iter, sol.res, sol.status,
            np.abs(sol.s).max() if len(sol.s) else 0.0, mr.L,
            # strict-complementarity margin: min over the inequality rows of max(multiplier, slack).  An interior-point solution
            # leaves lam_i t_i ~ 1e-11 on every row, so a weakly active bound (lam_i ~ 1e-5) sits t_i ~ 1e-6 off its bound and u0*
            # inherits that: below ~1e-4 the 1e-6 parity bar is not defined by the problem (SURVEY.md §8c "strict-complementarity filter")
            float(np.minimum(np.maximum(sol.lam, sol.t), 1e30).min()) if len(sol.lam) else 1e30]


def solve_set(name, kw, X0, U0=None, thetas=None, gamma=None, procs=6):
    jobs = [(name, kw, i, x0, None if thetas is None else thetas[i], None if U0 is None else U0[i], gamma) for i, x0 in enumerate(X0)]
    if procs > 1 and len(jobs) > 2:
        with mp.get_context("spawn").Pool(procs) as pool:   # fresh interpreters: forked children of a process that already ran torch hang
            rows = pool.map(_one, jobs, chunksize=1)
    else:
        rows = [_one(j) for j in jobs]
    return {k: np.array([r[c] for r in rows]) for c, k in enumerate(KEYS)}


def g2():
    for gamma, tag in ((0.99, "g099"), (0.9, "g09")):
        X0 = np.array([[0.5, 0.5], [0.2, 0.2], [0.7, -0.3], [0.3, 0.45]])
        r = solve_set("linear", {"gamma": gamma}, X0, procs=1)
        rq = solve_set("linear", {"gamma": gamma}, X0[1:3], U0=np.array([[-0.5], [0.25]]), procs=1)
        np.savez(os.path.join(HERE, f"g2_linear_{tag}.npz"), gamma=gamma, x0=X0, **r,
                 **{"q_" + k: v for k, v in rq.items()}, q_x0=X0[1:3], q_u0fix=np.array([[-0.5], [0.25]]))


def g3():
    """64 states of the reference's reset distribution + 64 near-upright states, seed 0, each at the nominal model
    parameters and at theta x {0.9, 1.1}; Q-mode point of scripts/cartpole_mpc_sensitivities.py:80-81."""
    P = make_cartpole()
    rng = np.random.default_rng(0)
    n = 64
    X = np.zeros((2 * n, 4))
    X[:n, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, n)
    X[n:] = rng.uniform(-1, 1, (n, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    scales = np.array([1.0, 0.9, 1.1])
    X0 = np.repeat(X, 3, axis=0)
    thetas = np.tile(P.p0, (len(X0), 1))
    thetas[:, :3] *= np.tile(scales, len(X))[:, None]
    r = solve_set("cartpole", {}, X0, thetas=thetas)
    xq = np.array([[0.0, 0.0, np.pi / 2, 0.0]])
    rq = solve_set("cartpole", {}, xq, U0=np.array([[-30.0]]), procs=1)
    keep = ("u0", "V", "dV", "dpi", "sqp_iter", "ipm_iter", "res", "status", "smax", "L", "sc")
    np.savez_compressed(os.path.join(HERE, "g3_cartpole.npz"), x0=X0, theta_model=thetas[:, :3], theta_scale=np.tile(scales, len(X)),
                        **{k: r[k] for k in keep}, X=r["X"][:12], U=r["U"][:12], PI=r["PI"][:12],
                        **{"q_" + k: v for k, v in rq.items()}, q_x0=xq, q_u0fix=np.array([[-30.0]]))


def g4():
    """All 10 points of the sweep of tests/test_chain_mass.py (C_3_0 over linspace(0.5, 1.5, 10) x nominal)."""
    P = make_chain_mass()
    p_idx = P.p_labels.index("C_3_0")
    vals = np.linspace(0.5 * P.p0[p_idx], 1.5 * P.p0[p_idx], 10)
    thetas = np.tile(P.p0, (len(vals), 1))
    thetas[:, p_idx] = vals
    X0 = np.tile(P.x0_default, (len(vals), 1))
    r = solve_set("chain", {}, X0, thetas=thetas, procs=5)
    keep = ("u0", "V", "dV", "dpi", "sqp_iter", "ipm_iter", "res", "status", "smax", "L", "sc")
    np.savez_compressed(os.path.join(HERE, "g4_chain5.npz"), x0=X0, p_idx=p_idx, p_vals=vals, x_ss=P.extra["x_ss"],
                        **{k: r[k] for k in keep}, X=r["X"][:2], U=r["U"][:2], PI=r["PI"][:2])


def g5():
    """chain_mass n_mass = 7 (nx = 33): the nominal initial state and two with the N(0, 1e-2) velocity perturbation of SURVEY.md §8d
    config 4 (seed 0)."""
    P = make_chain_mass(n_mass=7)
    rng = np.random.default_rng(0)
    X0 = np.tile(P.x0_default, (3, 1))
    M_ = 5
    X0[1:, 3 * (M_ + 1):] += rng.normal(0.0, 1e-2, (2, 3 * M_))
    r = solve_set("chain", {"n_mass": 7}, X0, procs=3)
    keep = ("u0", "V", "dV", "dpi", "sqp_iter", "ipm_iter", "res", "status", "smax", "L", "sc")
    np.savez_compressed(os.path.join(HERE, "g5_chain7.npz"), x0=X0, x_ss=P.extra["x_ss"], **{k: r[k] for k in keep}, X=r["X"][:1],
                        U=r["U"][:1], PI=r["PI"][:1])


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5"]
    for w in which:
        globals()[w]()
