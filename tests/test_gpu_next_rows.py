"""GPU tests of the callers of the hot path: batched Q-learning, the TD3/DPG actor, iterate store/load, RTI closed loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_qlearning_linear_system_two_episodes(oracle_port):
    """Config 1 (plumbing): the loop of rlmpc/examples/linear_system_mpc_qlearning.py on the batched engine; the first
    parameter step is checked against the same formula evaluated with the CPU oracle port on the recorded samples."""
    from mpc4rl_amd import BatchedLinearSystemEnv, BatchedQLearning, linear_system_ocp
    from oracle.problems import make_linear_system
    ocp = linear_system_ocp(discount_factor=0.9)
    env = BatchedLinearSystemEnv(8, device="cuda", lb_noise=-0.02, ub_noise=0.02, seed=0)
    ql = BatchedQLearning(ocp, env, episode_length=12, lr=1e-4, gamma=0.9)
    theta0 = ql.theta.clone()
    st = ql.run_episode()
    assert st.converged_fraction == 1.0 and np.isfinite(st.total_cost) and np.isfinite(st.td_error_mean)
    assert torch.allclose(ql.theta, theta0 + st.step) and float(st.step.abs().max()) > 0.0
    assert torch.equal(ql.rollout_mpc.get_theta(), ql.theta)
    # oracle check of one Q / V / dQ evaluation on fresh samples with the updated parameters
    P = make_linear_system(gamma=0.9)
    s = np.array([[0.5, 0.5], [0.3, -0.2]])
    a = np.array([[-0.3], [0.4]])
    th = ql.theta.cpu().numpy()
    from mpc4rl_amd import MPCBatch
    m = MPCBatch(ocp, 2)
    m.set_theta(ql.theta)
    rq = m.solve(s, u0=a, sens_v=True, cold=True)
    ref = oracle_port.solve(P, s, p=th, u0fix=a, gamma=0.9)
    assert np.allclose(rq.V.cpu().numpy(), ref.V, rtol=1e-9) and np.allclose(rq.dV_dp.cpu().numpy(), ref.dV, rtol=1e-7, atol=1e-9)
    st2 = ql.run_episode()
    assert np.isfinite(st2.td_error_mean)


def test_mpc_actor_forward_and_dpg_step():
    """Actor.forward == per-observation get_action loop of rlmpc/td3/policies.py:186-213, in one launch."""
    from mpc4rl_amd import CartpoleMPC, ContinuousCritic, MPCActor, cartpole_ocp
    torch.manual_seed(0)
    B = 12
    obs = (torch.rand(B, 4, dtype=torch.float64) * 2 - 1) * torch.tensor([0.5, 1.0, 0.3, 1.0], dtype=torch.float64)
    obs = obs.cuda()
    actor = MPCActor(cartpole_ocp(), B, device="cuda:0")
    act = actor(obs.float())
    assert act.shape == (B, 1) and act.dtype == torch.float32 and float(act.abs().max()) <= 1.0
    single = CartpoleMPC()
    for i in (0, 5, 11):
        single.reset(obs[i].cpu().numpy())
        a_i = single.get_action(obs[i].float().double().cpu().numpy())
        assert abs(a_i[0] - float(act[i, 0])) < 1e-5
    critic = ContinuousCritic(4, 1).cuda()
    theta0 = actor.parameters().clone()
    step = actor.dpg_step(obs.float(), critic, lr=1e-3)
    assert step.shape == theta0.shape and torch.isfinite(step).all()
    assert float(step[:3].abs().max()) > 0.0 and float(step[3:].abs().max()) == 0.0   # only (M, m, l) enter the policy
    assert torch.allclose(actor.parameters(), theta0 + step)


def test_store_load_iterate_and_rti_closed_loop(tmp_path):
    """store/load iterate (examples/chain_mass.py:119-120) and real-time iterations in closed loop: one QP per step from
    the stored iterate tracks the converged solution."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, MPCBatch, cartpole_ocp, load_iterate, store_iterate
    B = 32
    ocp = cartpole_ocp()
    env = BatchedCartPoleSwingUpEnv(B, device="cuda", seed=1)
    obs = env.reset()
    mpc = MPCBatch(ocp, B)
    r = mpc.solve(obs, cold=True)
    assert bool((r.status == 0).all())
    f = str(tmp_path / "iterate.pt")
    store_iterate(mpc, f)
    other = MPCBatch(ocp, B)
    load_iterate(other, f)
    r2 = other.solve(obs)                       # starts at the solution: converged at iteration 0
    assert bool((r2.status == 0).all()) and int(r2.iters[:, 0].max()) == 0
    assert torch.allclose(r2.u0, r.u0) and torch.allclose(r2.V, r.V)
    full = MPCBatch(ocp, B)
    for _ in range(5):
        a = 2.0 * ((r.u0 - (-30.0)) / 60.0) - 1.0
        obs, _, _, _ = env.step(a)
        r = mpc.solve(obs, rti=True)            # one QP from the previous (unshifted) iterate
        rf = full.solve(obs, cold=True)
        assert int(r.iters[:, 0].max()) <= 1
        assert float((r.u0 - rf.u0).abs().max()) < 0.5 and float(((r.V - rf.V).abs() / rf.V.abs()).max()) < 5e-2   # one QP per step: approximate by design


def test_weighted_grad_sum_kernel():
    """K5: the reduction kernel against torch, for the few-column (cartpole) and many-column (chain) shapes, strided rows."""
    from mpc4rl_amd.distributed import allreduce_weighted_grad
    torch.manual_seed(0)
    for B, n_p, n in ((4096, 83, 3), (1024, 499, 499), (7, 12, 12), (300, 20, 17)):
        full = torch.randn(B, n_p, dtype=torch.float64, device="cuda")
        w = torch.randn(B, dtype=torch.float64, device="cuda")
        g = full[:, :n]
        s, ws, cnt = allreduce_weighted_grad(g, w)
        ref = (w[:, None] * g).sum(0)
        assert cnt == B and torch.allclose(s, ref, rtol=1e-12, atol=1e-12) and abs(float(ws) - float(w.sum())) < 1e-10
