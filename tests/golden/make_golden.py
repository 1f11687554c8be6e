"""Generates the committed golden fixtures in tests/golden/ (run in the dev container: `python tests/golden/make_golden.py`).

The reference itself cannot run here or on the GPU box (acados / CasADi are not vendored and not installable),
and it stores no expected outputs, so these vectors come from the build's own independent oracle:
  * G1  closed-form discounted LQR / DARE known answers (scipy), SURVEY.md §8c "G1"
  * G2  linear_system  (rlmpc/mpc/linear_system/acados.py, tests/test_linear_example.py params)
  * G3  cartpole N=20  (config/cartpole.yaml, reset distribution of continuous_cartpole/environment.py:178-180)
  * G4  chain_mass n_mass=5: the parameter sweep of tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174
each solved by oracle/sqp_dense.py (dense SQP + dense IPM) with sensitivities from oracle/nlp_mirror.py (torch autograd of
the residual R(z,p) of nlp.py:1214 + dense solve, nlp.py:1410-1424).  Inputs and expected outputs only — no reference source.
"""
import json
import os
import sys

import numpy as np

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


def solve_set(P, X0, U0=None, thetas=None, gamma=None):
    keys = ["u0", "V", "dV", "dpi", "X", "U", "PI", "sqp_iter", "ipm_iter", "res", "status", "smax"]
    out = {k: [] for k in keys}
    for i, x0 in enumerate(X0):
        p = None if thetas is None else thetas[i]
        u0 = None if U0 is None else U0[i]
        sol = S.solve(P, x0, p=p, u0fix=u0, gamma=gamma)
        mr = M.evaluate(P, sol, x0, p=p, u0fix=u0, gamma=gamma)
        M.assert_reference_consistency(P, sol, mr)
        for k, v in zip(keys, [sol.u[0], sol.cost, mr.dL_dp[0], mr.dpi_dp, sol.x, sol.u, sol.pi, sol.sqp_iter, sol.ipm_iter, sol.res,
                               sol.status, np.abs(sol.s).max() if len(sol.s) else 0.0]):
            out[k].append(v)
        print(P.name, i, sol.status, sol.sqp_iter, sol.ipm_iter, sol.u[0], sol.cost, flush=True)
    return {k: np.array(v) for k, v in out.items()}


def g2():
    for gamma, tag in ((0.99, "g099"), (0.9, "g09")):
        P = make_linear_system(gamma=gamma)
        X0 = np.array([[0.5, 0.5], [0.2, 0.2], [0.7, -0.3], [0.3, 0.45]])
        r = solve_set(P, X0)
        rq = solve_set(P, X0[1:3], U0=np.array([[-0.5], [0.25]]))
        np.savez(os.path.join(HERE, f"g2_linear_{tag}.npz"), gamma=gamma, x0=X0, **r,
                 **{"q_" + k: v for k, v in rq.items()}, q_x0=X0[1:3], q_u0fix=np.array([[-0.5], [0.25]]))


def g3():
    P = make_cartpole()
    rng = np.random.default_rng(0)
    n = 4
    X0 = np.zeros((2 * n, 4))
    X0[:n, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, n)
    X0[n:] = rng.uniform(-1, 1, (n, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    thetas = np.tile(P.p0, (2 * n, 1))
    thetas[1::2, :3] *= np.array([0.9, 1.1, 1.05])
    r = solve_set(P, X0, thetas=thetas)
    xq = np.array([[0.0, 0.0, np.pi / 2, 0.0]])          # scripts/cartpole_mpc_sensitivities.py:80-81
    rq = solve_set(P, xq, U0=np.array([[-30.0]]))
    np.savez(os.path.join(HERE, "g3_cartpole.npz"), x0=X0, theta=thetas, **r, **{"q_" + k: v for k, v in rq.items()}, q_x0=xq,
             q_u0fix=np.array([[-30.0]]))


def g4():
    P = make_chain_mass()
    p_idx = P.p_labels.index("C_3_0")
    vals = np.linspace(0.5 * P.p0[p_idx], 1.5 * P.p0[p_idx], 10)[[0, 4, 9]]   # 3 of the 10 points of tests/test_chain_mass.py
    thetas = np.tile(P.p0, (len(vals), 1))
    thetas[:, p_idx] = vals
    X0 = np.tile(P.x0_default, (len(vals), 1))
    r = solve_set(P, X0, thetas=thetas)
    np.savez_compressed(os.path.join(HERE, "g4_chain5.npz"), x0=X0, p_idx=p_idx, p_vals=vals, x_ss=P.extra["x_ss"],
                        **{k: v for k, v in r.items() if k not in ("dV", "dpi")}, dV=r["dV"].astype(np.float64),
                        dpi=r["dpi"].astype(np.float64))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4"]
    for w in which:
        globals()[w]()
