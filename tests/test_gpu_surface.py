"""The reference's own tests and usage patterns, run against the drop-in surface (mpc4rl_amd.MPC) on the GPU.

  * tests/test_linear_example.py:8-26      construction asserts
  * tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174   set_p / update / update_nlp / get_pi / get_dpi_dp sweep
  * rlmpc/examples/linear_system_mpc_qlearning.py:153-205   get_action / q_update / get_dQ_dp / get_Q / update / get_V
  * scripts/linear_system_mpc_nlp.py:17-106   FD of V and Q against get_dV_dp / get_dQ_dp
  * BASELINE config 3: cartpole du0*/dp and dV/dp against central finite differences on a 256-instance subset
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PARAM = {
    "A": np.array([[1.0, 0.25], [0.0, 1.0]]), "B": np.array([[0.03125], [0.25]]), "Q": np.identity(2), "R": np.identity(1),
    "b": np.array([[0.0], [0.0]]), "f": np.array([[0.0], [0.0], [0.0]]), "V_0": np.array([1e-3]),
}


def test_mpc_initialization():
    """tests/test_linear_example.py:8-26, same asserts."""
    from mpc4rl_amd import LinearSystemMPC as AcadosMPC
    mpc = AcadosMPC(PARAM, discount_factor=0.99)
    assert mpc is not None
    assert mpc.ocp_solver is not None
    assert mpc.ocp_solver.acados_ocp is not None
    assert mpc.ocp_solver.acados_ocp.model is not None
    assert mpc.ocp_solver.acados_ocp.dims is not None
    assert mpc.ocp_solver.acados_ocp.cost is not None
    assert mpc.ocp_solver.acados_ocp.constraints
    assert mpc.get_parameter_labels() == ["A_0", "A_1", "A_2", "A_3", "B_0", "B_1", "b_0", "b_1", "V_0", "f_0", "f_1", "f_2"]
    assert mpc.get_p().shape == (12,) and mpc.discount_factor == 0.99


def test_chain_mass_main_nlp():
    """rlmpc/examples/chain_mass.py:133-174 with np_test = 10 (what tests/test_chain_mass.py runs)."""
    from mpc4rl_amd import ChainMassMPC
    mpc = ChainMassMPC(discount_factor=1.0)
    x0 = mpc.ocp.x0
    p_idx = mpc.ocp.p_labels.index("C_3_0")
    p_nom = mpc.nlp.p.val.cat.full().flatten()
    p_var = np.linspace(0.5 * p_nom[p_idx], 1.5 * p_nom[p_idx], 10)
    u_opt, sens_u = [], []
    for i in range(10):
        p = p_nom.copy()
        p[p_idx] = p_var[i]
        mpc.set_p(p)
        _ = mpc.update(x0)
        mpc.update_nlp()
        u_opt.append(mpc.get_pi())
        sens_u.append(mpc.get_dpi_dp()[:, p_idx].flatten())
        assert mpc.nlp.assert_kkt_residual(tol=1e-5)      # the implicit check of update_nlp (nlp.py:1445-1537), chain tol
    u_opt, sens_u = np.vstack(u_opt), np.vstack(sens_u)
    assert u_opt.shape == (10, 3) and sens_u.shape == (10, 3)
    sens_fd = np.gradient(u_opt, p_var, axis=0)               # plot_results, chain_mass.py:28-36
    assert np.abs(sens_fd[1:-1] - sens_u[1:-1]).max() < 5e-3 * max(1.0, np.abs(sens_u).max())


def test_qlearning_call_pattern_and_errors():
    """One learning sweep of rlmpc/examples/linear_system_mpc_qlearning.py:153-205 on a short synthetic episode."""
    from mpc4rl_amd import LinearSystemMPC
    mpc = LinearSystemMPC(PARAM, discount_factor=0.9)
    rng = np.random.default_rng(0)
    A, B = np.array([[0.9, 0.35], [0.0, 1.1]]), np.array([[0.0813], [0.2]])   # env plant, linear_system/environment.py:15
    obs = np.array([0.5, 0.5])
    mpc.reset(obs)
    traj = []
    for _ in range(6):
        action = mpc.get_action(obs)
        assert action.shape == (1,) and mpc.status == 0
        nxt = A @ obs + B @ action + np.array([rng.uniform(-0.02, 0.02), 0.0])
        cost = 0.5 * obs @ obs + 0.5 * action @ action
        traj.append((obs, action, cost))
        obs = nxt
    n = len(traj)
    dQ_dp, q, v = np.zeros((n, 12)), np.zeros(n), np.zeros(n)
    mpc.reset(traj[0][0])
    for i, (o, a, _) in enumerate(traj):
        status = mpc.q_update(o, mpc.unscale_action(mpc.scale_action(a)))
        assert status == 0
        dQ_dp[i, :] = mpc.get_dQ_dp()
        q[i] = mpc.get_Q()
        mpc.update(o)
        v[i] = mpc.get_V()
    assert np.all(np.isfinite(dQ_dp)) and np.all(q >= v - 1e-9)       # Q(s,a) >= V(s) = min_a Q(s,a)
    cost = np.array([c for _, _, c in traj])
    td = cost[:-1] + 0.9 * v[1:] - q[:-1]
    dp = np.mean(np.vstack([1e-4 * td[i] * dQ_dp[i, :] for i in range(n - 1)]), axis=0)
    mpc.set_parameter(mpc.get_parameter_values() + dp)
    assert np.allclose(mpc.get_p(), mpc.ocp.p0 + dp)
    # error behaviour: update / q_update raise RuntimeError on status != 0 (mpc.py:81-83,197-198); get_action stores it
    from mpc4rl_amd import CartpoleMPC, cartpole_ocp
    cp = CartpoleMPC(cartpole_ocp(max_iter=2))
    with pytest.raises(RuntimeError):
        cp.update(np.array([0.0, 0.0, 3.14, 0.0]))
    cp.reset(np.array([0.0, 0.0, 3.14, 0.0]))
    a = cp.get_action(np.array([0.0, 0.0, 3.14, 0.0]))
    assert cp.status == 2 and -1.0 <= a[0] <= 1.0                      # scaled action (cartpole/acados.py:239-249)


def test_value_gradients_vs_finite_differences_linear():
    """scripts/linear_system_mpc_nlp.py:17-106: np.gradient of V (and Q) over a +-10 % parameter line vs get_dV_dp / get_dQ_dp
    (reference tolerance atol = 1e-1; here 1e-3 relative)."""
    from mpc4rl_amd import LinearSystemMPC
    mpc = LinearSystemMPC(PARAM)
    mpc.set_discount_factor(0.99)
    x0, u0 = np.array([0.2, 0.2]), np.array([-0.5])
    p_nom = mpc.get_parameters().copy()
    for i_param in range(12):
        if p_nom[i_param] == 0.0:
            continue
        line = np.linspace(0.9 * p_nom[i_param], 1.1 * p_nom[i_param], 21)
        for qmode in (False, True):
            V, dV = [], []
            for val in line:
                p = p_nom.copy()
                p[i_param] = val
                mpc.set_parameter(p)
                if qmode:
                    mpc.q_update(x0, u0)
                    V.append(mpc.get_Q()), dV.append(mpc.get_dQ_dp()[0, i_param])
                else:
                    mpc.update(x0)
                    V.append(mpc.get_V()), dV.append(mpc.get_dV_dp()[0, i_param])
            true = np.gradient(np.array(V), line[1] - line[0])[1:-1]
            assert np.allclose(true, np.array(dV)[1:-1], rtol=1e-3, atol=1e-3), (i_param, qmode)
    mpc.set_parameter(p_nom)


def test_cartpole_sensitivities_vs_finite_differences_batch():
    """BASELINE config 3: validate dV/dp and du0*/dp on a 256-instance subset against central finite differences
    (delta = 1e-5 relative, tolerance 1e-4 relative)."""
    from mpc4rl_amd import MPCBatch, cartpole_ocp
    B = 256
    rng = np.random.default_rng(5)
    x0 = np.zeros((B, 4))
    x0[: B // 2, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B // 2)
    x0[B // 2:] = rng.uniform(-1, 1, (B // 2, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    ocp = cartpole_ocp(tol=1e-10)
    mpc = MPCBatch(ocp, B)
    r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True)
    ok = (r.status == 0).cpu().numpy()
    assert ok.mean() > 0.95
    dV, dpi = r.dV_dp.cpu().numpy(), r.dpi_dp.cpu().numpy()
    fdV, fdu = np.zeros((B, 3)), np.zeros((B, 3))
    for i in range(3):
        d = 1e-5 * ocp.p0[i]
        outs = []
        for sgn in (+1, -1):
            th = ocp.p0.copy()
            th[i] += sgn * d
            mpc.set_theta(torch.as_tensor(th))
            rr = mpc.solve(x0, cold=True)
            ok &= (rr.status == 0).cpu().numpy()
            outs.append((rr.V.cpu().numpy(), rr.u0.cpu().numpy()[:, 0]))
        fdV[:, i] = (outs[0][0] - outs[1][0]) / (2 * d)
        fdu[:, i] = (outs[0][1] - outs[1][1]) / (2 * d)
    eV = np.abs(dV[ok, :3] - fdV[ok]) / np.maximum(np.abs(fdV[ok]), 1.0)
    eu = np.abs(dpi[ok, 0, :3] - fdu[ok]) / np.maximum(np.abs(fdu[ok]), 1.0)
    print("FD check: dV max rel", eV.max(), "dpi max rel", eu.max(), "instances", int(ok.sum()))
    assert eV.max() < 1e-4 and eu.max() < 1e-4
    assert np.all(dV[:, 3:] == 0.0) and np.all(dpi[:, :, 3:] == 0.0)    # W / yref entries: zero gradient (nlp.py:1039-1055)


def test_cartpole_cost_through_set_parameter_and_cost_set(oracle_port):
    """MPC.set_parameter / ocp_solver.cost_set push W_0, W, yref_0, yref into the solver (mpc.py:233-257): the next solve uses them.
    Compared with the oracle port solving with the same parameter vector."""
    from mpc4rl_amd import CartpoleMPC
    from oracle.problems import make_cartpole
    P = make_cartpole()
    mpc = CartpoleMPC()
    x0 = np.array([0.2, -0.5, 0.25, 0.4])
    mpc.update(x0)
    u_nom, V_nom = mpc.get_pi().copy(), mpc.get_V()
    ps = mpc.nlp.p.sym(mpc.get_p())                       # the struct view the reference builds in set / set_parameter
    assert ps.keys() == ["model", "W_0", "W", "W_e", "yref_0", "yref", "yref_e"]
    assert ps["model"].full().shape == (3, 1) and ps["W"].full().shape == (5, 5) and ps["yref_e"].full().shape == (4, 1)
    W = np.diag([5.0, 0.2, 20.0, 0.1, 0.02])
    W[0, 2] = W[2, 0] = 0.5
    ps["W"], ps["W_0"] = W, 2 * W
    ps["yref"] = np.array([0.1, 0.0, 0.05, 0.0, 0.5])
    mpc.set_parameter(ps.cat.full().flatten())
    assert np.allclose(mpc.nlp.get_parameter("W").full(), W) and np.allclose(mpc.nlp.get_parameter("W_0").full(), 2 * W)
    mpc.update(x0)
    ref = oracle_port.solve(P, x0[None], p=mpc.get_p()[None])
    assert abs(mpc.get_pi()[0] - u_nom[0]) > 1.0 and abs(mpc.get_V() - V_nom) > 1e-2
    assert abs(mpc.get_pi()[0] - ref.u0[0, 0]) < 1e-6 * max(1.0, abs(ref.u0[0, 0])) and abs(mpc.get_V() - ref.V[0]) < 1e-8
    assert np.allclose(mpc.get_dV_dp(), ref.dV, rtol=1e-6, atol=1e-9) and np.allclose(mpc.get_dpi_dp(), ref.dpi[0], rtol=1e-6, atol=1e-8)
    assert np.all(mpc.get_dV_dp()[0, 3:] == 0.0)
    # cost_set(stage, "yref", ..) on an interior stage = the shared yref; on stage 0 = yref_0
    mpc.ocp_solver.cost_set(3, "yref", np.zeros(5))
    mpc.ocp_solver.cost_set(0, "yref", np.array([0.2, 0.0, 0.0, 0.0, 0.0]))
    mpc.update(x0)
    ref = oracle_port.solve(P, x0[None], p=mpc.get_p()[None])
    assert np.allclose(mpc.get_p()[74:79], 0.0) and mpc.get_p()[69] == 0.2
    assert abs(mpc.get_pi()[0] - ref.u0[0, 0]) < 1e-6 * max(1.0, abs(ref.u0[0, 0])) and abs(mpc.get_V() - ref.V[0]) < 1e-8


def test_set_before_first_solve_and_after_reset(oracle_port):
    """The reference initialises with `for stage: ocp_solver.set(stage, "x", x0)` (mpc.py:208-210) before any solve, and may call
    set() after reset(): both must be a cold start from that guess — never a warm start from uninitialised or stale multipliers."""
    from mpc4rl_amd import CartpoleMPC
    from oracle.problems import make_cartpole
    P = make_cartpole()
    x0 = np.array([0.0, 0.0, 3.0, 0.0])
    ref = oracle_port.solve(P, x0[None])
    mpc = CartpoleMPC()
    x, u, pi, bnd, _ = mpc._batch.get_iterate()           # a new handle holds the cold iterate, not garbage
    assert float(x.abs().max()) == 0.0 and float(u.abs().max()) == 0.0 and float(bnd[:, :2].abs().max()) == 0.0
    assert bool((bnd[:, 2:4] == 1.0).all())
    for stage in range(mpc.ocp.N + 1):
        mpc.ocp_solver.set(stage, "x", x0)
    assert np.allclose(mpc.get(5, "x"), x0)
    mpc.update(x0)
    assert mpc.status == 0 and abs(mpc.get_pi()[0] - ref.u0[0, 0]) < 1e-6 * abs(ref.u0[0, 0]) and abs(mpc.get_V() - ref.V[0]) < 1e-8
    # after reset the stored iterate is the cold one again; a set() on top of it must not resurrect the old multipliers
    mpc.reset(x0)
    assert np.allclose(mpc.get(7, "x"), x0) and np.all(mpc.get(7, "lam") == 0.0) and np.all(mpc.get(7, "t") == 1.0)
    mpc.set(2, "u", np.array([1.5]))
    assert np.all(mpc.get(7, "lam") == 0.0) and mpc.get(2, "u")[0] == 1.5
    mpc.update(x0)
    assert mpc.status == 0 and abs(mpc.get_V() - ref.V[0]) < 1e-8
    # acados-order multipliers of an interior stage: lbu, lbx(4), ubu, ubx(4) (common/utils.py:4-25) against the oracle's rows
    lam, t = mpc.get(1, "lam"), mpc.get(1, "t")
    assert lam.shape == (10,) and t.shape == (10,)
    b = ref.BND[0]                                         # [10, N+1, nu+nx]: lam_l, lam_u, t_l, t_u, ...
    assert np.allclose(lam, np.concatenate([b[0, 1], b[1, 1]]), atol=1e-7) and np.allclose(t, np.concatenate([b[2, 1], b[3, 1]]), rtol=1e-6)


def test_constraints_set_get_L_and_nlp_set(oracle_port):
    """constraints_set(0, "lbu"/"ubu", u0) pins u_0 exactly as q_update does (mpc.py:71-76, 87-88); tighter input bounds reach the
    solve; nlp.vars.val / nlp.set mirror writes (mpc.py:67-68, 75-76, 191-192); get_L = float(nlp.L.val) (mpc.py:325-332)."""
    from mpc4rl_amd import CartpoleMPC, LinearSystemMPC
    from oracle.problems import make_cartpole
    from oracle import nlp_mirror, sqp_dense
    P = make_cartpole()
    mpc = CartpoleMPC()
    x0, u0 = np.array([0.1, 0.0, 0.3, 0.0]), np.array([-4.0])
    mpc.q_update(x0, u0)
    Q, dQ = mpc.get_Q(), mpc.get_dQ_dp().copy()
    # the same through the solver shim, the way MPC.q_update is written in the reference
    mpc.reset(x0)
    mpc.nlp.set(0, "lbx", x0)
    mpc.nlp.vars.val["ubx_0"] = x0
    mpc.ocp_solver.constraints_set(0, "lbu", u0)
    mpc.ocp_solver.constraints_set(0, "ubu", u0)
    assert mpc.ocp_solver.solve() == 0
    assert abs(mpc.ocp_solver.get_cost() - Q) < 1e-9 and np.allclose(mpc.ocp_solver.get(0, "u"), u0)
    mpc.ocp_solver.constraints_set(0, "lbu", mpc.ocp_solver.acados_ocp.constraints.lbu)
    mpc.ocp_solver.constraints_set(0, "ubu", mpc.ocp_solver.acados_ocp.constraints.ubu)
    assert mpc.ocp_solver.solve() == 0
    ref = oracle_port.solve(P, x0[None])
    assert abs(mpc.get_pi()[0] - ref.u0[0, 0]) < 1e-6 * max(1.0, abs(ref.u0[0, 0]))
    # tighter input bounds on every stage: the oracle with the same bounds
    P2 = make_cartpole()
    P2.lbu[:], P2.ubu[:] = -5.0, 5.0
    x1 = np.array([0.0, 0.0, 3.0, 0.0])
    for stage in range(mpc.ocp.N):
        mpc.ocp_solver.constraints_set(stage, "lbu", np.array([-5.0]))
        mpc.ocp_solver.constraints_set(stage, "ubu", np.array([5.0]))
    assert np.allclose(mpc.nlp.vars.val["lbu_3"], -5.0) and np.allclose(mpc.nlp.vars.val["ubu_0"], 5.0)
    mpc.reset(x1)
    mpc.get_action(x1)
    ref2 = oracle_port.solve(P2, x1[None])
    assert mpc.status == ref2.status[0]
    if mpc.status == 0:
        assert abs(mpc.get_V() - ref2.V[0]) < 1e-6 * max(1.0, abs(ref2.V[0]))
    u_all = np.array([mpc.get(k, "u")[0] for k in range(mpc.ocp.N)])
    assert np.abs(u_all).max() <= 5.0 + 1e-9 and np.allclose(u_all, ref2.U[0, :, 0], atol=1e-5)
    # Lagrangian: the mirror's L at the oracle's solution (linear system, soft bound rows included)
    from oracle.problems import make_linear_system
    Pl = make_linear_system(gamma=0.99)
    lm = LinearSystemMPC(discount_factor=0.99)
    for xl in (np.array([0.5, 0.5]), np.array([0.7, -0.3])):
        lm.reset(xl)
        lm.update(xl)
        sol = sqp_dense.solve(Pl, xl)
        mr = nlp_mirror.evaluate(Pl, sol, xl)
        assert abs(lm.get_L() - mr.L) < 1e-7 * max(1.0, abs(mr.L)) and abs(lm.nlp.L.val - lm.get_L()) == 0.0
        assert abs(lm.get_L() - lm.get_V()) < 1e-5          # lam'h + pi'g vanish at a KKT point up to the barrier parameter
    assert mpc.nlp.vars.val["gamma"] == 1.0 and mpc.nlp.vars.val["dT", 0] == mpc.ocp.dT
    assert np.allclose(mpc.nlp.vars.val["x", 0], x1)


def test_chain_sizes_the_library_accepts():
    """n_mass is a free integer in the reference (chain_mass/ocp_utils.py:344-350); the library instantiates 3 .. 7 and refuses the
    rest at construction (MPCRL_E_MODEL), never at solve time."""
    from mpc4rl_amd import MPCBatch, chain_mass_ocp
    for n_mass in (3, 4, 5, 6, 7):
        mpc = MPCBatch(chain_mass_ocp(n_mass=n_mass), 2)
        assert mpc.n_p == chain_mass_ocp(n_mass=n_mass).n_p
        del mpc
    for n_mass in (8, 9):
        with pytest.raises(RuntimeError, match="mpcrl_create failed"):
            MPCBatch(chain_mass_ocp(n_mass=n_mass), 2)


def test_chain_horizons_the_library_accepts(oracle_port):
    """The chain kernels stage whole trajectories in LDS (the output reduction: (1 + nu) x ((N + 1) nx + N nu) doubles): a horizon whose
    requests do not fit one workgroup's 64 KB is refused by mpcrl_create (MPCRL_E_ARG), never at solve time after the SQP has run.
    n_mass 7 (nx 33): N = 55 is the longest horizon; it solves with sensitivities and agrees with the port; N = 56 is refused.
    n_mass 5 (nx 21) runs up to the 63 stages a wavefront has lanes for."""
    from mpc4rl_amd import MPCBatch, chain_mass_ocp
    from oracle.problems import make_chain_mass
    with pytest.raises(RuntimeError, match="mpcrl_create failed"):
        MPCBatch(chain_mass_ocp(n_mass=7, N=56), 2)
    rng = np.random.default_rng(4)
    for n_mass, N in ((7, 55), (5, 63)):
        ocp = chain_mass_ocp(n_mass=n_mass, N=N)
        P = make_chain_mass(n_mass=n_mass, N=N)
        M = n_mass - 2
        x0 = np.tile(ocp.x0, (4, 1))
        x0[:, 3 * (M + 1):] += rng.normal(0.0, 1e-2, (4, 3 * M))
        r = MPCBatch(ocp, 4).solve(x0, sens_v=True, sens_pi=True, cold=True)
        ref = oracle_port.solve(P, x0)
        assert bool((r.status == 0).all()) and np.all(ref.status == 0)
        for a, b in ((r.u0, ref.u0), (r.V, ref.V), (r.dV_dp, ref.dV), (r.dpi_dp, ref.dpi)):
            a, b = a.cpu().numpy().reshape(4, -1), np.asarray(b).reshape(4, -1)
            assert (np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-300)).max() < 1e-6
