"""CPU tests of the oracle (test infrastructure): golden vectors, known answers, FD checks, KKT thresholds.

The reference stores no golden numbers and cannot run here, so the oracle is pinned by
(i) the closed-form LQR/DARE known answers G1, (ii) finite differences, (iii) the KKT-consistency thresholds the
reference applies to every solver output (rlmpc/mpc/nlp.py:1445-1537), (iv) agreement between two independent
implementations: dense Python + torch autograd (sqp_dense / nlp_mirror) vs the structured C++ port."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b, floor=1.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def test_g1_lqr_dare_known_answers(oracle_port):
    """With all bounds inactive the OCP is a discounted LQR: u0* and V have closed forms."""
    from oracle.problems import make_linear_system
    g1 = json.load(open(os.path.join(GOLD, "g1_lqr.json")))
    assert np.allclose(g1["P_dare"], [[7.464124567610356, 4.031128874149254], [4.031128874149254, 7.518320906916593]], rtol=1e-12)
    # literal values quoted in SURVEY.md §8c
    for c, (u_ref, v_ref) in zip(g1["cases"], [(-1.15469289, 2.8815879029), (-1.12639085, 2.8021094924), (-0.88166796, 2.1607881519)]):
        assert abs(c["u0"] - u_ref) < 1e-8 and abs(c["V"] - v_ref) < 1e-9
    for c in g1["cases"]:
        P = make_linear_system(gamma=c["gamma"])
        P.lbu[:], P.ubu[:], P.lbx[:], P.ubx[:] = -100, 100, -100, 100
        r = oracle_port.solve(P, g1["x0"])
        assert r.status[0] == 0
        assert abs(r.u0[0, 0] - c["u0"]) < 1e-9 and abs(r.V[0] - c["V"]) < 1e-9


@pytest.mark.parametrize("tag", ["g099", "g09"])
def test_port_vs_golden_linear(oracle_port, tag):
    from oracle.problems import make_linear_system
    g = np.load(os.path.join(GOLD, f"g2_linear_{tag}.npz"))
    P = make_linear_system(gamma=float(g["gamma"]))
    r = oracle_port.solve(P, g["x0"])
    assert np.all(r.status == 0) and np.abs(r.sqp_iter - g["sqp_iter"]).max() <= 1 and np.abs(r.ipm_iter - g["ipm_iter"]).max() <= 3
    assert rel(r.u0, g["u0"]) < 1e-9 and rel(r.V, g["V"]) < 1e-9 and rel(r.X, g["X"]) < 1e-9 and rel(r.PI, g["PI"]) < 1e-8
    assert rel(r.dV, g["dV"]) < 1e-8
    strict = g["smax"] < 1e-9                  # du0/dp is ill-posed where a soft bound is active (quirk q1)
    assert rel(r.dpi[strict], g["dpi"][strict]) < 1e-6
    q = oracle_port.solve(P, g["q_x0"], u0fix=g["q_u0fix"])
    assert rel(q.V, g["q_V"]) < 1e-9 and rel(q.dV, g["q_dV"]) < 1e-8 and np.all(q.dpi == 0.0)
    assert np.abs(g["q_dpi"]).max() < 1e-6     # the mirror's own value: ~0 (u_0 pinned by lbu_0 = ubu_0)


def test_port_vs_golden_cartpole(oracle_port):
    from oracle.problems import make_cartpole
    g = np.load(os.path.join(GOLD, "g3_cartpole.npz"))
    P = make_cartpole()
    r = oracle_port.solve(P, g["x0"], p=g["theta"])
    # iteration counts may differ by one where a stopping test is met to within rounding (dense KKT solves vs Riccati)
    assert np.all(r.status == 0) and np.abs(r.sqp_iter - g["sqp_iter"]).max() <= 1 and np.abs(r.ipm_iter - g["ipm_iter"]).max() <= 3
    # where the counts differ the two runs stop at different iterates, both within tol = 1e-6 of the KKT point: the bar is 1e-6
    for k, a in (("u0", r.u0), ("V", r.V), ("X", r.X), ("U", r.U), ("PI", r.PI), ("dV", r.dV)):
        assert rel(a, g[k]) < 1e-6, k
    assert rel(r.dpi, g["dpi"]) < 1e-6
    assert np.all(r.res < P.tol)               # assert_kkt_residual (nlp.py:1295-1299)
    q = oracle_port.solve(P, g["q_x0"], u0fix=g["q_u0fix"])
    assert rel(q.V, g["q_V"]) < 1e-9 and rel(q.dV, g["q_dV"]) < 1e-8


def test_port_vs_golden_chain(oracle_port):
    """The sweep of tests/test_chain_mass.py (C_3_0 over [0.05, 0.15]): u0* and du0*/dp[:, p_idx]."""
    from oracle.problems import make_chain_mass
    g = np.load(os.path.join(GOLD, "g4_chain5.npz"))
    P = make_chain_mass()
    assert rel(P.extra["x_ss"], g["x_ss"]) < 1e-12
    theta = np.tile(P.p0, (len(g["p_vals"]), 1))
    theta[:, int(g["p_idx"])] = g["p_vals"]
    r = oracle_port.solve(P, g["x0"], p=theta)
    assert np.all(r.status == 0) and np.abs(r.sqp_iter - g["sqp_iter"]).max() <= 1
    assert rel(r.u0, g["u0"]) < 1e-8 and rel(r.V, g["V"]) < 1e-9 and rel(r.dV, g["dV"]) < 1e-7
    assert rel(r.dpi, g["dpi"], floor=np.abs(g["dpi"]).max()) < 1e-6


def test_dense_python_vs_port_cartpole(oracle_port):
    """Two implementations of the same iteration (dense KKT solves vs Riccati recursion) agree to rounding."""
    from oracle import nlp_mirror, sqp_dense
    from oracle.problems import make_cartpole
    P = make_cartpole()
    x0 = np.array([0.3, 0.5, 0.2, -0.4])
    sol = sqp_dense.solve(P, x0)
    mr = nlp_mirror.evaluate(P, sol, x0)
    nlp_mirror.assert_reference_consistency(P, sol, mr)        # thresholds of nlp.py:1445-1537
    assert np.abs(mr.R).max() < 1e-6                          # assert_kkt_residual
    r = oracle_port.solve(P, x0)
    assert abs(r.sqp_iter[0] - sol.sqp_iter) <= 1 and abs(r.ipm_iter[0] - sol.ipm_iter) <= 3
    # both stop at an NLP residual < 1e-6 after inexactly solved intermediate QPs: the iterates agree to the parity bar (1e-6), not to rounding
    assert rel(r.X[0], sol.x) < 1e-6 and rel(r.U[0], sol.u) < 1e-6 and rel(r.PI[0], sol.pi) < 1e-6
    assert rel(r.dV[0], mr.dL_dp[0]) < 1e-6 and rel(r.dpi[0], mr.dpi_dp) < 1e-6


def test_value_gradient_vs_finite_differences_linear(oracle_port):
    """Pattern of scripts/linear_system_mpc_nlp.py:17-106 (central differences of V and Q), at rtol 1e-4 instead of atol 1e-1."""
    from oracle.problems import make_linear_system
    P = make_linear_system(gamma=0.99)
    for x0, u0 in (([0.2, 0.2], None), ([0.2, 0.2], [-0.5]), ([0.5, 0.5], None)):
        r = oracle_port.solve(P, x0, u0fix=u0)
        d = 1e-4
        theta = np.tile(P.p0, (2 * P.n_p, 1))
        for i in range(P.n_p):
            theta[2 * i, i] += d
            theta[2 * i + 1, i] -= d
        X0 = np.tile(x0, (2 * P.n_p, 1))
        U0 = None if u0 is None else np.tile(u0, (2 * P.n_p, 1))
        f = oracle_port.solve(P, X0, p=theta, u0fix=U0, flags=0, tol=1e-10)
        fd = (f.V[0::2] - f.V[1::2]) / (2 * d)
        assert rel(r.dV[0], fd) < 1e-4      # FD noise floor: IPM complementarity 1e-11 * 318 rows / (2 d)


def test_policy_gradient_vs_finite_differences_cartpole(oracle_port):
    """du0*/dp against central differences of u0* (the check behind scripts/cartpole_mpc_sensitivities.py)."""
    from oracle.problems import make_cartpole
    P = make_cartpole()
    x0 = np.array([0.3, 0.5, 0.2, -0.4])
    r = oracle_port.solve(P, x0, tol=1e-10)
    d = 1e-6
    theta = np.tile(P.p0, (6, 1))
    for i in range(3):
        theta[2 * i, i] += d
        theta[2 * i + 1, i] -= d
    f = oracle_port.solve(P, np.tile(x0, (6, 1)), p=theta, flags=0, tol=1e-11)
    fd_u = (f.u0[0::2, 0] - f.u0[1::2, 0]) / (2 * d)
    fd_v = (f.V[0::2] - f.V[1::2]) / (2 * d)
    assert rel(r.dpi[0, 0, :3], fd_u, floor=np.abs(fd_u).max()) < 1e-5
    assert rel(r.dV[0, :3], fd_v) < 1e-5
    assert np.all(r.dV[0, 3:] == 0.0) and np.all(r.dpi[0, 0, 3:] == 0.0)   # non-parameterised NLS cost (nlp.py:1039-1055)


def test_edge_cases(oracle_port):
    """max-iter status, warm start from the solution (0 iterations), batch of one, per-instance theta."""
    from oracle.problems import make_cartpole
    P = make_cartpole()
    x0 = np.array([[0.0, 0.0, 3.14, 0.0]])
    r = oracle_port.solve(P, x0, max_iter=2)
    assert r.status[0] == 2 and r.sqp_iter[0] == 2
    r = oracle_port.solve(P, x0)
    w = oracle_port.solve(P, x0, warm=r)
    assert w.status[0] == 0 and w.sqp_iter[0] == 0 and rel(w.V, r.V) < 1e-14 and rel(w.dV, r.dV) < 1e-12
