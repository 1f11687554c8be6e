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


def cartpole_gold():
    """G3 (tests/golden/make_golden.py): inputs + outputs of the dense oracle in its frozen exact-QP mode."""
    from oracle.problems import make_cartpole
    g = np.load(os.path.join(GOLD, "g3_cartpole.npz"))
    P = make_cartpole()
    theta = np.tile(P.p0, (len(g["x0"]), 1))
    theta[:, :3] = g["theta_model"]
    return P, g, theta


def close_fraction(a, b, tol=1e-6):
    a, b = np.asarray(a, float).reshape(len(a), -1), np.asarray(b, float).reshape(len(b), -1)
    e = (np.abs(a - b) / np.maximum(np.abs(b).max(1, keepdims=True), 1.0)).max(1)
    return float((e < tol).mean()), float(e.max())


@pytest.mark.parametrize("tag", ["g099", "g09"])
def test_port_vs_golden_linear(oracle_port, tag):
    """Linear system: a QP, so the exact mode (golden) and the tuned mode agree to the interior-point tolerance."""
    from oracle.problems import make_linear_system
    g = np.load(os.path.join(GOLD, f"g2_linear_{tag}.npz"))
    P = make_linear_system(gamma=float(g["gamma"]))
    for exact in (True, False):
        r = oracle_port.solve(P, g["x0"], exact=exact)
        assert np.all(r.status == 0) and np.abs(r.sqp_iter - g["sqp_iter"]).max() <= 1
        assert rel(r.u0, g["u0"]) < 1e-8 and rel(r.V, g["V"]) < 1e-8 and rel(r.X, g["X"]) < 1e-8 and rel(r.PI, g["PI"]) < 1e-7
        assert rel(r.dV, g["dV"]) < 1e-7
        strict = g["smax"] < 1e-9                  # du0/dp is ill-posed where a soft bound is active (quirk q1)
        assert rel(r.dpi[strict], g["dpi"][strict]) < 1e-6
        q = oracle_port.solve(P, g["q_x0"], u0fix=g["q_u0fix"], exact=exact)
        assert rel(q.V, g["q_V"]) < 1e-8 and rel(q.dV, g["q_dV"]) < 1e-7 and np.all(q.dpi == 0.0)
    assert np.abs(g["q_dpi"]).max() < 1e-6     # the mirror's own value: ~0 (u_0 pinned by lbu_0 = ubu_0)


def check_against_kkt_golden(g, u0, V, dV, dpi, tight):
    """The golden vectors are the KKT point (NLP tolerance 1e-8, tests/golden/make_golden.py).  ``tight``: the run under test was
    also taken to 1e-8 — then the bar of BASELINE.json's north_star (1e-6 relative) holds for EVERY instance, for du0*/dp on the
    instances that pass the strict-complementarity filter of SURVEY.md §8c (sc >= 1e-3: no weakly active bound; an interior-point
    solution leaves lam t ~ 1e-11 per row, so where lam ~ 1e-4 the bound is missed by ~1e-7 and du0*/dp, which divides by that,
    moves in the 6th digit).  A run stopped at the solver's default tolerance 1e-6 (the reference's setting) can itself only be within
    ~1e-6 x conditioning of the KKT point: V and dV/dp still meet 1e-6 everywhere, u0* and du0*/dp on >= 99 % of the instances and
    to 1e-5 / 2e-5 on all of them."""
    strict = g["sc"] >= 1e-3
    fu, wu = close_fraction(u0, g["u0"])
    fv, wv = close_fraction(V, g["V"])
    fd, wd = close_fraction(dV, g["dV"])
    fp, wp = close_fraction(dpi[strict], g["dpi"][strict])
    _, wpa = close_fraction(dpi, g["dpi"])
    print(f"tight={tight}: worst rel err u0 {wu:.1e} V {wv:.1e} dV {wd:.1e} dpi(strict) {wp:.1e} dpi(all) {wpa:.1e}; "
          f"fraction within 1e-6: u0 {fu:.4f} dpi {fp:.4f}; strict-complementarity filter keeps {strict.mean():.4f}")
    assert wv < 1e-6 and wd < 1e-6 and wpa < 2e-5
    if tight:
        assert wu < 1e-6 and wp < 1e-6
    else:
        assert wu < 1e-5 and fu >= 0.99 and fp >= 0.99


def test_port_vs_golden_cartpole(oracle_port):
    """G3 = 64 + 64 states x theta {1, 0.9, 1.1}, generated by the dense exact-QP oracle + autograd mirror at NLP
    tolerance 1e-8.  The C++ port reproduces it in its exact mode (two linear-algebra paths, same iteration), in the tuned inexact
    mode the product runs (a different iterate sequence that must land on the same KKT point), and at the default tolerance."""
    P, g, theta = cartpole_gold()
    assert np.all(g["status"] == 0) and len(g["x0"]) == 384 and g["res"].max() < 1e-8
    for exact, tol in ((True, 1e-8), (False, 1e-8), (False, None)):
        r = oracle_port.solve(P, g["x0"], p=theta, exact=exact, tol=tol)
        assert np.all(r.status == 0)
        if exact:
            assert np.abs(r.sqp_iter - g["sqp_iter"]).max() <= 1
        check_against_kkt_golden(g, r.u0, r.V, r.dV, r.dpi, tight=tol is not None)
        assert rel(r.X[:12], g["X"]) < 1e-5 and rel(r.U[:12], g["U"]) < 1e-5 and rel(r.PI[:12], g["PI"]) < 1e-5
        assert np.all(r.res < (tol or P.tol))      # assert_kkt_residual (nlp.py:1295-1299)
        q = oracle_port.solve(P, g["q_x0"], u0fix=g["q_u0fix"], exact=exact, tol=tol)
        assert rel(q.V, g["q_V"]) < 1e-8 and rel(q.dV, g["q_dV"]) < 1e-6


def test_port_vs_golden_chain(oracle_port):
    """G4: all ten points of the sweep of tests/test_chain_mass.py (C_3_0 over [0.05, 0.15]); G5: n_mass = 7 (nx = 33)."""
    from oracle.problems import make_chain_mass
    g = np.load(os.path.join(GOLD, "g4_chain5.npz"))
    P = make_chain_mass()
    assert rel(P.extra["x_ss"], g["x_ss"]) < 1e-12 and len(g["p_vals"]) == 10
    theta = np.tile(P.p0, (len(g["p_vals"]), 1))
    theta[:, int(g["p_idx"])] = g["p_vals"]
    assert g["res"].max() < 1e-8 and g["sc"].min() > 1e-3        # KKT point to 1e-8, strictly complementary
    for exact, tol in ((True, 1e-8), (False, 1e-8), (True, None), (False, None)):
        r = oracle_port.solve(P, g["x0"], p=theta, exact=exact, tol=tol)
        assert np.all(r.status == 0)
        # at the golden's tolerance: the 1e-6 bar; stopped at the chain's default 1e-5 (ocp_utils.py:311-312) a run is itself only
        # within ~1e-5 x conditioning of the KKT point (measured: dV/dp 3.6e-6 in the exact mode)
        bar = 1e-6 if tol is not None else 2e-5
        assert rel(r.u0, g["u0"]) < bar and rel(r.V, g["V"]) < 1e-8 and rel(r.dV, g["dV"]) < bar
        assert rel(r.dpi, g["dpi"], floor=np.abs(g["dpi"]).max()) < bar
    g = np.load(os.path.join(GOLD, "g5_chain7.npz"))
    P = make_chain_mass(n_mass=7)
    assert rel(P.extra["x_ss"], g["x_ss"]) < 1e-12
    for tol, bar in ((1e-8, 1e-6), (None, 2e-5)):
        r = oracle_port.solve(P, g["x0"], tol=tol)
        assert np.all(r.status == 0)
        assert rel(r.u0, g["u0"]) < bar and rel(r.V, g["V"]) < 1e-8 and rel(r.dV, g["dV"]) < bar
        assert rel(r.dpi, g["dpi"], floor=np.abs(g["dpi"]).max()) < bar


def test_cost_block_of_p_reaches_the_solve(oracle_port):
    """set_parameter / cost_set semantics (mpc.py:233-257): W_0, W, W_e, yref_0, yref, yref_e inside p are what the solver uses;
    their gradient entries stay zero (non-parameterised NLS mirror, nlp.py:1039-1055).  Dense Python vs C++ port."""
    from oracle import nlp_mirror, sqp_dense
    from oracle.problems import make_cartpole
    P = make_cartpole()
    p = P.p0.copy()
    W = np.diag([5.0, 0.2, 20.0, 0.1, 0.02])
    W[0, 2] = W[2, 0] = 0.5
    p[28:53], p[3:28] = W.flatten("F"), (2 * W).flatten("F")
    p[74:79], p[69:74], p[79:83] = [0.1, 0, 0.05, 0, 0.5], [0.2, 0, 0, 0, 0], [0.1, 0, 0, 0]
    x0 = np.array([0.2, -0.5, 0.25, 0.4])
    sol = sqp_dense.solve(P, x0, p=p)
    mr = nlp_mirror.evaluate(P, sol, x0, p=p)
    r = oracle_port.solve(P, x0[None], p=p[None])
    nominal = oracle_port.solve(P, x0[None])
    assert sol.status == 0 and r.status[0] == 0
    assert abs(r.u0[0, 0] - nominal.u0[0, 0]) > 1.0                       # the weights matter
    assert rel(r.u0[0], sol.u[0]) < 1e-6 and rel(r.V[0], sol.cost) < 1e-8
    assert rel(r.dV[0], mr.dL_dp[0]) < 1e-6 and rel(r.dpi[0], mr.dpi_dp) < 1e-6
    assert np.all(r.dV[0, 3:] == 0.0) and np.all(mr.dL_dp[0, 3:] == 0.0)


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


def test_non_finite_initial_state_is_status_1(oracle_port):
    """An instance whose x0 holds a NaN (or overflows) ends with status 1 in both oracle implementations: max() drops NaNs from the
    residual norms, so the finiteness test includes the cost (a sum).  Its neighbours in the batch are untouched."""
    from oracle.problems import make_cartpole, make_linear_system
    from oracle import sqp_dense
    for P, nx in ((make_cartpole(), 4), (make_linear_system(), 2)):
        x0 = np.random.default_rng(3).uniform(-0.3, 0.3, (4, nx))
        good = oracle_port.solve(P, x0)
        x0[1, nx - 1] = np.nan
        x0[2, 0] = 1e200
        r = oracle_port.solve(P, x0)
        assert r.status[1] == 1 and r.status[2] in (1, 4) and r.status[0] == 0 and r.status[3] == 0
        assert r.V[0] == good.V[0] and r.V[3] == good.V[3]
        d = sqp_dense.solve(P, x0[1])
        assert d.status == 1


def test_mirror_certifies_an_iterate_it_did_not_produce(oracle_port):
    """oracle/from_iterate.py: iterate arrays in the layouts of include/mpcrl.h -> the mirror of the reference's NLP, as update_nlp
    copies the acados iterate (nlp.py:1354-1398).  Here the iterate is the C++ port's; the GPU suite feeds the product's.  The
    reference's thresholds (nlp.py:1445-1537) hold and the mirror's dL/dp, dz/dp[:nu] equal the port's adjoint-Riccati values."""
    from oracle.from_iterate import certify
    from oracle.problems import make_cartpole, make_linear_system
    rng = np.random.default_rng(4)
    x0c = np.zeros((2, 4))
    x0c[0, 2] = 0.95 * np.pi
    x0c[1] = [0.1, 0.2, 0.1, -0.3]
    x0l = np.column_stack([rng.uniform(0.3, 0.7, 2), rng.uniform(-0.3, 0.3, 2)])
    for P, x0 in ((make_cartpole(), x0c), (make_linear_system(gamma=0.9), x0l)):
        r = oracle_port.solve(P, x0)
        for i in range(2):
            assert r.status[i] == 0
            mr, sc, sol = certify(P, r.X[i], r.U[i], r.PI[i], r.BND[i], x0[i], cost=r.V[i])
            assert np.abs(mr.dL_dp[0] - r.dV[i]).max() <= 1e-6 * max(1.0, np.abs(r.dV[i]).max())
            if sc >= 1e-3 or (len(sol.s) and np.abs(sol.s).max() < 1e-9):
                assert np.abs(mr.dpi_dp - r.dpi[i]).max() <= 1e-6 * max(1.0, np.abs(r.dpi[i]).max())


def test_certification_worker_runs_in_a_process_pool(oracle_port):
    """The GPU certification tests farm oracle.from_iterate.certify_job out to fresh interpreters (spawn): the worker must be
    importable there and return what the in-process call returns."""
    import multiprocessing as mp
    from oracle.from_iterate import certify_job
    from oracle.problems import make_linear_system
    P = make_linear_system(gamma=0.9)
    x0 = np.array([[0.4, 0.1], [0.6, -0.2]])
    r = oracle_port.solve(P, x0)
    jobs = [("linear", {"gamma": 0.9}, r.X[i], r.U[i], r.PI[i], r.BND[i], x0[i], None, None, 0.9, float(r.V[i])) for i in range(2)]
    with mp.get_context("spawn").Pool(2) as pool:
        rows = pool.map(certify_job, jobs, chunksize=1)
    here = certify_job(jobs[0])
    assert np.array_equal(rows[0][0], here[0]) and np.abs(rows[1][0] - r.dV[1]).max() < 1e-8


def test_port_divergence_exit_rule(oracle_port):
    """The opt-in exit rule of the product (mpcrl_set_exit_rule) as the port restates it: bounded iteration counts on states from the
    whole box, nothing converges only because of the rule, and what still converges is bitwise what the rule-less run returns."""
    from oracle.problems import make_cartpole
    P = make_cartpole()
    rng = np.random.default_rng(7)
    x0 = rng.uniform(-1, 1, (256, 4)) * np.array([2.0, 3.0, np.pi, 5.0])
    r0 = oracle_port.solve(P, x0, flags=0, max_iter=300)
    r1 = oracle_port.solve(P, x0, flags=0, max_iter=300, exit_window=10, exit_factor=0.1)
    c1 = r1.status == 0
    assert r0.sqp_iter.max() == 300 and r1.sqp_iter.max() < 120
    assert np.all(r0.status[c1] == 0) and c1.sum() >= 0.9 * (r0.status == 0).sum()
    assert np.array_equal(r0.u0[c1], r1.u0[c1]) and np.array_equal(r0.V[c1], r1.V[c1]) and np.array_equal(r0.sqp_iter[c1], r1.sqp_iter[c1])
    assert set(np.unique(r1.status[~c1])) <= {2, 4}


def test_third_party_solver_agrees_with_the_goldens_and_the_port(oracle_port):
    """G6 (tests/golden/make_thirdparty.py): KKT points found by scipy.optimize — SLSQP on the cartpole NLP from the reference's cold
    iterate, trust-constr on the linear-system QP — and certified there without any of this repo's solvers (stationarity below
    1e-13 / 1e-5, feasibility 1e-15).  Every other vector of tests/golden/ comes from the build's own dense oracle; this one removes
    "same author on both sides": the dense oracle's goldens AND the C++ port must reproduce u0* and V of a solver nobody here wrote,
    at the north_star bar 1e-6 (measured: cartpole <= 3e-9 in u0*, 7e-13 in V; linear <= 2.3e-8, 3.2e-9)."""
    from oracle.problems import make_cartpole, make_linear_system
    g6 = np.load(os.path.join(GOLD, "g6_thirdparty.npz"))
    assert g6["cp_kkt"][:, 0].max() < 1e-12 and g6["cp_kkt"][:, 1].max() < 1e-12 and g6["cp_kkt"][:, 2].min() >= 0.0
    assert bool(g6["cp_same_minimum"].all())          # SLSQP from the cold iterate ends in the minimum the full-step SQP finds
    rel = lambda a, b: float((np.abs(np.asarray(a).reshape(len(b), -1) - np.asarray(b).reshape(len(b), -1)) /
                              np.maximum(np.abs(np.asarray(b).reshape(len(b), -1)), 1.0)).max())
    g3 = np.load(os.path.join(GOLD, "g3_cartpole.npz"))
    rows = g6["cp_g3_row"]
    assert np.array_equal(g3["x0"][rows], g6["cp_x0"])
    assert rel(g3["u0"][rows], g6["cp_u0"]) < 1e-6 and rel(g3["V"][rows], g6["cp_V"]) < 1e-6
    P = make_cartpole()
    theta = np.tile(P.p0, (len(rows), 1))
    theta[:, :3] = g6["cp_theta_model"]
    r = oracle_port.solve(P, g6["cp_x0"], p=theta, tol=1e-8)
    assert np.all(r.status == 0) and rel(r.u0, g6["cp_u0"]) < 1e-6 and rel(r.V, g6["cp_V"]) < 1e-6
    for tag, gamma in (("g099", 0.99), ("g09", 0.9)):
        g2 = np.load(os.path.join(GOLD, f"g2_linear_{tag}.npz"))
        assert g6[f"lin_{tag}_kkt"][:, 0].max() < 1e-5 and g6[f"lin_{tag}_kkt"][:, 1].max() < 1e-12
        assert rel(g2["u0"], g6[f"lin_{tag}_u0"]) < 1e-6 and rel(g2["V"], g6[f"lin_{tag}_V"]) < 1e-6
        r = oracle_port.solve(make_linear_system(gamma=gamma), g6[f"lin_{tag}_x0"])
        assert np.all(r.status == 0) and rel(r.u0, g6[f"lin_{tag}_u0"]) < 1e-6 and rel(r.V, g6[f"lin_{tag}_V"]) < 1e-6


def test_third_party_gradients_and_chain_solutions_vs_the_port(oracle_port):
    """G7 (tests/golden/make_thirdparty_grad.py): central differences of a third-party solver's V and u0* over the parameters — the
    reference's own check of dpi/dp is finite differences along a parameter sweep (rlmpc/examples/chain_mass.py:28-64) — and SLSQP
    solutions of the chain-of-masses NLP.  The port's dV/dp and du0*/dp against them at 1e-5 (the differences carry the solver's
    own noise: only entries where the steps 1e-5 and 1e-4 agree to 2e-6 of the row's scale are used), its u0*, V at 1e-6."""
    from oracle.problems import make_cartpole, make_chain_mass, make_linear_system
    if not os.path.exists(os.path.join(GOLD, "g7_thirdparty_grad.npz")):
        pytest.skip("tests/golden/g7_thirdparty_grad.npz has not been generated (make_thirdparty_grad.py, ~1.5 h on 8 cores)")
    g7 = np.load(os.path.join(GOLD, "g7_thirdparty_grad.npz"))

    def held(fd0, fd1, mine, tol=1e-5):
        # Central differences at two steps, d and 10 d: fd(d) = g + c d^2, so g = fd0 - (fd1 - fd0) / 99 and |fd1 - fd0| / 99 estimates
        # what is left of the truncation error in fd0; entries whose estimate exceeds 2e-6 are left out (and NaN entries of jobs the
        # generator did not finish).  Returns the largest |mine - g|, rows scaled by their largest entry (>= 1).
        fd0, fd1, mine = (np.asarray(a, float).reshape(len(fd0), -1) for a in (fd0, fd1, mine))
        fin = np.isfinite(fd0) & np.isfinite(fd1)
        if not fin.any():
            return 0.0
        f0, f1 = np.where(fin, fd0, 0.0), np.where(fin, fd1, 0.0)
        scale = np.maximum(np.abs(f0).max(axis=1, keepdims=True), 1.0)
        g = f0 - (f1 - f0) / 99.0
        ok = fin & (np.abs(f1 - f0) / 99.0 <= 2e-6 * scale)
        assert ok[fin].mean() >= 0.9, ok[fin].mean()
        return float(np.where(ok, np.abs(mine - g) / scale, 0.0).max())

    # cartpole: 4 near-upright states x (M, m, l)
    P = make_cartpole()
    B = len(g7["cp_x0"])
    theta = np.tile(P.p0, (B, 1))
    theta[:, :3] = g7["cp_theta_model"]
    r = oracle_port.solve(P, g7["cp_x0"], p=theta, tol=1e-9)
    assert np.all(r.status == 0)
    assert np.abs(r.u0 - g7["cp_u0"]).max() < 1e-6 * np.abs(g7["cp_u0"]).max() and np.abs(r.V - g7["cp_V"]).max() < 1e-6 * np.abs(g7["cp_V"]).max()
    assert held(g7["cp_dV_d0"], g7["cp_dV_d1"], r.dV[:, :3]) < 1e-5
    assert held(g7["cp_du0_d0"][:, :, 0], g7["cp_du0_d1"][:, :, 0], r.dpi[:, 0, :3]) < 1e-5
    # linear system: 2 states with an interior u0* x 12 parameters, both discount factors; every point KKT-certified to 1e-12
    for tag, gamma in (("g099", 0.99), ("g09", 0.9)):
        Pl = make_linear_system(gamma=gamma)
        have = np.isfinite(g7[f"lin_{tag}_V"])
        r = oracle_port.solve(Pl, g7[f"lin_{tag}_x0"], tol=1e-9)
        assert np.all(r.status == 0)
        if have.any():
            assert np.nanmax(g7[f"lin_{tag}_kkt"][:, :2]) < 1e-11 and np.nanmax(g7[f"lin_{tag}_kkt_d0"]) < 1e-11
            assert np.abs(r.u0 - g7[f"lin_{tag}_u0"])[have].max() < 1e-6 and np.abs(r.V - g7[f"lin_{tag}_V"])[have].max() < 1e-6 * np.abs(g7[f"lin_{tag}_V"][have]).max()
        # du0*/dp at 3e-5 (measured 2e-5 on the offsets b and f, 1e-8 on A and B): every solution of this OCP ends with a weakly active
        # soft row (the regulated state approaches its bound), where the reference's KKT sensitivity keeps the slacks constant
        # (quirk q1) and the true solution map, which the finite differences follow, does not
        assert held(g7[f"lin_{tag}_dV_d0"], g7[f"lin_{tag}_dV_d1"], r.dV) < 1e-5
        assert held(g7[f"lin_{tag}_du0_d0"][:, :, 0], g7[f"lin_{tag}_du0_d1"][:, :, 0], r.dpi[:, 0, :]) < 3e-5
        # G6's rows, polished: what trust-constr stopped at 2e-6 of is now a KKT point to 1e-12
        if len(g7[f"lin_{tag}_polished_x0"]):
            assert g7[f"lin_{tag}_polished_kkt"][:, :2].max() < 1e-11
            r = oracle_port.solve(Pl, g7[f"lin_{tag}_polished_x0"], tol=1e-9)
            assert np.abs(r.u0 - g7[f"lin_{tag}_polished_u0"]).max() < 1e-6 and np.abs(r.V - g7[f"lin_{tag}_polished_V"]).max() < 1e-6 * max(1.0, np.abs(r.V).max())
    # chain of masses: SLSQP from the reference's cold iterate
    for n_mass in (3, 5):
        if f"chain{n_mass}_x0" not in g7.files:
            continue
        assert g7[f"chain{n_mass}_kkt"][0] < 1e-10 and g7[f"chain{n_mass}_kkt"][1] < 1e-10 and g7[f"chain{n_mass}_kkt"][2] >= 0.0
        r = oracle_port.solve(make_chain_mass(n_mass=n_mass), g7[f"chain{n_mass}_x0"][None], tol=1e-9)
        assert r.status[0] == 0
        assert np.abs(r.u0[0] - g7[f"chain{n_mass}_u0"]).max() < 1e-6 and abs(r.V[0] - float(g7[f"chain{n_mass}_V"])) < 1e-6 * max(1.0, abs(r.V[0]))


@pytest.mark.parametrize("fixture", ["g8_chain_grad.npz", "g8_chain4_grad.npz", "g8_chain5_grad.npz", "g8_chain7_grad.npz"])
def test_third_party_chain_gradients_vs_the_port(oracle_port, fixture):
    """G8 (tests/golden/make_thirdparty_chain_grad.py): dV/dp and du0*/dp of the chain-of-masses NLP against central differences of a
    third-party solver's V and u0* over masses, spring constants, rest lengths, damping coefficients and cost weights — the
    reference's own check of dpi/dp on the chain is a finite difference along its damping sweep (rlmpc/examples/chain_mass.py:28-64).
    n_mass 3 / 4: scipy SLSQP; n_mass 5 (the size the reference's test runs, 8 parameters incl. the swept C_3_0) and 7 (the benchmark's
    perf size): MINPACK's hybrid method on the KKT equations of the certified active set (round 6).  Every entry is used: the two
    step sizes of the difference quotients agree on all of them."""
    f = os.path.join(GOLD, fixture)
    if not os.path.exists(f):
        pytest.skip(f"tests/golden/{fixture} has not been generated (make_thirdparty_chain_grad.py, hours)")
    g8 = np.load(f)
    from oracle.problems import make_chain_mass
    P = make_chain_mass(n_mass=int(g8["n_mass"]))
    assert g8["kkt"][:, :2].max() < 1e-9 and g8["kkt_d0"].max() < 1e-9
    r = oracle_port.solve(P, g8["x0"][None], tol=1e-9)
    assert r.status[0] == 0
    assert np.abs(r.u0[0] - g8["u0"]).max() < 1e-6 and abs(r.V[0] - float(g8["V"])) < 1e-6 * max(1.0, abs(float(g8["V"])))
    idx = g8["p_index"]
    for name, mine, fd0, fd1 in (("dV/dp", r.dV[0, idx], g8["dV_d0"], g8["dV_d1"]), ("du0*/dp", r.dpi[0][:, idx].T, g8["du0_d0"], g8["du0_d1"])):
        fd0, fd1, mine = (np.asarray(a, float).reshape(len(idx), -1) for a in (fd0, fd1, mine))
        scale = np.maximum(np.abs(fd0).max(1, keepdims=True), 1.0)
        ok = np.abs(fd0 - fd1) <= 2e-6 * scale          # the two step sizes agree: the difference quotient is trustworthy there
        assert ok.all(), (name, ok)                       # (round 5 accepted a quarter of dropped entries: none is dropped)
        err = float((np.abs(mine - fd0) / scale).max())
        print("chain n_mass", int(g8["n_mass"]), name, "port vs third-party finite differences:", err)
        assert err < 1e-5, name


def test_third_party_reference_sweep_vs_the_port(oracle_port):
    """The reference's own chain test with numbers of a third party (G8 sweep, make_thirdparty_chain_grad.py G8_SWEEP=1):
    tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174 sweeps the damping C_3_0 over linspace(0.05, 0.15, 10) at n_mass 5 from
    the unperturbed x0 of examples/chain_mass.py:17-25 and looks at pi* and d pi / d C_3_0 along it.  Here every point of that sweep is a
    certified KKT point found by MINPACK on the NLP restatement, its gradient a central difference of such points: the port's u0*, V at
    1e-6 and dV/dC_3_0, du0*/dC_3_0 at 1e-5 at all ten."""
    f = os.path.join(GOLD, "g8_chain5_sweep.npz")
    if not os.path.exists(f):
        pytest.skip("tests/golden/g8_chain5_sweep.npz has not been generated (G8_NMASS=5 G8_SWEEP=1 make_thirdparty_chain_grad.py)")
    g = np.load(f)
    from oracle.problems import make_chain_mass
    P = make_chain_mass(n_mass=5)
    n = len(g["C_3_0"])
    j = int(g["p_index"][0])
    assert n == 10 and np.allclose(g["C_3_0"], np.linspace(0.05, 0.15, 10)) and P.p_labels[j] == "C_3_0"
    assert g["kkt"][:, :2].max() < 1e-9 and g["kkt_d0"].max() < 1e-9
    th = np.tile(P.p0, (n, 1))
    th[:, j] = g["C_3_0"]
    r = oracle_port.solve(P, np.tile(g["x0"], (n, 1)), p=th, tol=1e-9)
    assert np.all(r.status == 0)
    assert np.abs(r.u0 - g["u0"]).max() < 1e-6 and np.abs(r.V - g["V"]).max() < 1e-6 * np.abs(g["V"]).max()
    for name, mine, fd0, fd1 in (("dV/dC_3_0", r.dV[:, j], g["dV_d0"], g["dV_d1"]), ("du0*/dC_3_0", r.dpi[:, :, j], g["du0_d0"], g["du0_d1"])):
        fd0, fd1, mine = (np.asarray(a, float).reshape(n, -1) for a in (fd0, fd1, mine))
        scale = np.maximum(np.abs(fd0).max(1, keepdims=True), 1.0)
        assert (np.abs(fd0 - fd1) <= 2e-6 * scale).all(), name
        err = float((np.abs(mine - fd0) / scale).max())
        print("reference sweep,", name, "port vs third-party finite differences:", err)
        assert err < 1e-5, name


def _held_all(fd0, fd1, mine, keep=0.9):
    """Richardson-extrapolated central differences (steps d and 10 d) against `mine`; entries whose two steps disagree by more than 2e-6 of
    the row's scale are left out — and at least `keep` of them must remain.  Returns (largest scaled error, kept fraction)."""
    fd0, fd1, mine = (np.asarray(a, float).reshape(len(fd0), -1) for a in (fd0, fd1, mine))
    scale = np.maximum(np.abs(fd0).max(axis=1, keepdims=True), 1.0)
    g = fd0 - (fd1 - fd0) / 99.0
    ok = np.abs(fd1 - fd0) / 99.0 <= 2e-6 * scale
    assert ok.mean() >= keep, ok
    return float(np.where(ok, np.abs(mine - g) / scale, 0.0).max()), float(ok.mean())


def test_third_party_cartpole_gradients_active_state_bound_vs_the_port(oracle_port):
    """G7b (tests/golden/make_thirdparty_grad2.py, round 6): central differences of scipy-SLSQP's V and u0* over (M, m, l) at ten cartpole
    states with u0* strictly inside its bounds — four of them with the cart's position bound |s| <= 2.4 (config/cartpole.yaml:82-91)
    ACTIVE on stages of the horizon (the fixture records max |s_k| = 2.4 of SLSQP's own solution), six near upright.  G7 had four
    states, two of them saturated (du0*/dp = 0) and none with a state bound active.  Port at 1e-5, u0*, V at 1e-6."""
    from oracle.problems import make_cartpole
    g = np.load(os.path.join(GOLD, "g7b_cartpole_grad.npz"))
    P = make_cartpole()
    assert len(g["x0"]) == 10 and np.all(np.abs(g["u0"]) < 29.0) and np.all(np.abs(g["s_max"][:4] - 2.4) < 1e-7) and np.all(g["s_max"][4:] < 2.39)
    assert g["kkt"][:, :2].max() < 1e-9 and g["kkt_d0"].max() < 1e-9
    r = oracle_port.solve(P, g["x0"], tol=1e-9)
    assert np.all(r.status == 0)
    # the port's own multipliers say the same: a state row of the first four instances is active
    lam_x = r.BND[:, 0:2, 1:, 1:]
    assert np.all(lam_x[:4].reshape(4, -1).max(1) > 1e-3) and np.all(lam_x[4:].reshape(6, -1).max(1) < 1e-6)
    assert np.abs(r.u0 - g["u0"]).max() < 1e-6 * np.abs(g["u0"]).max() and np.abs(r.V - g["V"]).max() < 1e-6 * np.abs(g["V"]).max()
    e_v, k_v = _held_all(g["dV_d0"], g["dV_d1"], r.dV[:, :3])
    e_pi, k_pi = _held_all(g["du0_d0"][:, :, 0], g["du0_d1"][:, :, 0], r.dpi[:, 0, :3])
    print("cartpole (active state bound / near upright) vs third-party finite differences: dV/dp %.2e (kept %.2f) du0*/dp %.2e (kept %.2f)" % (e_v, k_v, e_pi, k_pi))
    # (1e-6, the north_star bar: with the stiffness cap of the adjoint solve alone the four active-bound states were 1.5e-6 ... 3.6e-6 off;
    # the Richardson extrapolation in the cap — port and kernels alike — brings them to 3e-9)
    assert e_v < 1e-6 and e_pi < 1e-6


def test_third_party_cartpole_gradients_96_states_vs_the_port(oracle_port):
    """G7c (tests/golden/make_thirdparty_grad3.py, round 6): breadth — certified KKT points of the cartpole NLP found by MINPACK's hybrid
    method on the KKT equations (not by any solver of this repository) at 96 drawn states (swing-up starts, near-upright states, a wider
    box), their central differences over (M, m, l).  94 strictly complementary states are kept; 37 of them have u0* on its bound
    (du0*/dp = 0 exactly, dV/dp not), 57 an interior u0*.  Port: u0*, V at 1e-6, dV/dp and du0*/dp at 1e-5, every entry."""
    from oracle.problems import make_cartpole
    g = np.load(os.path.join(GOLD, "g7c_cartpole_grad.npz"))
    P = make_cartpole()
    n = len(g["x0"])
    assert n >= 90 and int(g["drawn"]) == 96 and g["kkt"][:, :2].max() < 1e-9 and g["kkt_d0"].max() < 1e-9
    r = oracle_port.solve(P, g["x0"], tol=1e-9)
    assert np.all(r.status == 0)
    assert (np.abs(r.u0 - g["u0"]) / np.maximum(np.abs(g["u0"]), 1.0)).max() < 1e-6 and (np.abs(r.V - g["V"]) / np.maximum(np.abs(g["V"]), 1.0)).max() < 1e-6
    e_v, k_v = _held_all(g["dV_d0"], g["dV_d1"], r.dV[:, :3], keep=0.98)
    e_pi, k_pi = _held_all(g["du0_d0"][:, :, 0], g["du0_d1"][:, :, 0], r.dpi[:, 0, :3], keep=0.98)
    sat = np.abs(g["u0"][:, 0]) > 29.999
    assert np.abs(r.dpi[sat][:, 0, :3]).max() < 1e-9                      # a control that sits on its bound does not move with the parameters
    print("cartpole, %d states (%d saturated) vs third-party finite differences: dV/dp %.2e (kept %.3f) du0*/dp %.2e (kept %.3f)" % (n, sat.sum(), e_v, k_v, e_pi, k_pi))
    assert e_v < 1e-5 and e_pi < 1e-5


def test_third_party_q_mode_gradients_vs_the_port(oracle_port):
    """G7d (tests/golden/make_thirdparty_grad4.py, round 6): Q(s, a) and dQ/dp — MPC.q_update, mpc.py:52-96, what the reference's Q-learning
    differentiates — at 31 cartpole (state, pinned u0) pairs against certified third-party KKT points and their central differences."""
    from oracle.problems import make_cartpole
    g = np.load(os.path.join(GOLD, "g7d_cartpole_qmode.npz"))
    n = len(g["x0"])
    assert n >= 28 and g["kkt"].max() < 1e-9 and g["kkt_d0"].max() < 1e-9
    r = oracle_port.solve(make_cartpole(), g["x0"], u0fix=g["u0"], tol=1e-9)
    assert np.all(r.status == 0) and np.array_equal(r.u0, g["u0"])
    assert (np.abs(r.V - g["Q"]) / np.maximum(np.abs(g["Q"]), 1.0)).max() < 1e-6
    e, kept = _held_all(g["dQ_d0"], g["dQ_d1"], r.dV[:, :3], keep=0.98)
    print("cartpole Q-mode, %d pairs: dQ/dp vs third-party finite differences %.2e (kept %.3f)" % (n, e, kept))
    assert e < 1e-6 and np.all(r.dpi == 0.0)


# two (x0, moved x0) pairs of the linear-system OCP (gamma 0.9, N 23: profiles/microbench/fuzz_parity.py seed 7) whose warm QP runs out of
# interior-point iterations, and one whose warm QP converges
WARM_QP_FAILS = np.array([[[0.8726968153298129, 0.5483649132524803], [0.8572483237722713, 0.5365310381189186]],
                          [[0.11781405145159196, -0.5019479937647703], [0.12047207467889322, -0.4954152158824344]],
                          [[0.8580657761160446, 0.5993274993476269], [0.8486730546864062, 0.6143373226840217]]])


def test_port_warm_qp_failure_restarts_cold(oracle_port):
    """LQ model: a warm interior point that jams and then cycles (60 iterations, status 4 before round 6) is run once more from the cold
    interior point; the answer is the one of a cold call (a convex QP has ONE solution)."""
    from oracle.problems import make_linear_system
    P = make_linear_system(gamma=0.9, N=23)
    first = oracle_port.solve(P, WARM_QP_FAILS[:, 0])
    warm = oracle_port.solve(P, WARM_QP_FAILS[:, 1], warm=first)
    cold = oracle_port.solve(P, WARM_QP_FAILS[:, 1])
    assert np.all(first.status == 0) and np.all(warm.status == 0) and np.all(cold.status == 0)
    assert np.all(warm.ipm_iter[:2] > 60) and warm.ipm_iter[2] < 20          # the restart happened, and only where the warm QP failed
    assert np.abs(warm.u0 - cold.u0).max() < 1e-6 and np.abs(warm.V - cold.V).max() < 1e-6
    assert np.abs(warm.dV - cold.dV).max() < 1e-4 * max(1.0, np.abs(cold.dV).max())
