"""GPU tests of the callers of the hot path: batched Q-learning, the TD3/DPG actor, iterate store/load, RTI closed loop."""
import os

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
    # the parameter step recomputed independently: Q(s_i, a_i), dQ/dp_i and V(s_i) of every recorded sample from the CPU oracle
    # port at the parameters the episode was run with, then the formula of linear_system_mpc_qlearning.py:193-205
    P = make_linear_system(gamma=0.9)
    S, A, C = [t.cpu().numpy() for t in ql.last_episode]            # [T, E, .]
    T, E = S.shape[:2]
    n = T - 1
    s, a = S[:n].reshape(n * E, -1), A[:n].reshape(n * E, -1)
    th0 = theta0.cpu().numpy()
    oq = oracle_port.solve(P, s, p=th0, u0fix=a, gamma=0.9)
    ov = oracle_port.solve(P, s, p=th0, gamma=0.9)
    assert np.all(oq.status == 0) and np.all(ov.status == 0)
    q, v, dq = oq.V.reshape(n, E), ov.V.reshape(n, E), oq.dV.reshape(n, E, -1)
    td = C[: n - 1] + 0.9 * v[1:] - q[:-1]
    step_ref = np.mean((1e-4 * td)[..., None] * dq[: n - 1], axis=(0, 1))
    assert np.allclose(st.step.cpu().numpy(), step_ref, rtol=1e-6, atol=1e-12)
    assert abs(st.td_error_mean - td.mean()) < 1e-8 * max(1.0, abs(td.mean()))
    # the rollout itself: a_t = u0*(s_t) of the oracle for the recorded states (warm-started full SQP lands on the same optimum)
    oa = oracle_port.solve(P, S[:, 0], p=th0, gamma=0.9, flags=0)
    assert np.allclose(A[:, 0], oa.u0, rtol=1e-6, atol=1e-6)      # two interior-point runs of one QP (complementarity 1e-11): ~3e-7
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


@pytest.mark.parametrize("name", ["cartpole", "linear", "chain"])
def test_rti_step_vs_oracle(oracle_port, name):
    """MPCRL_RTI = exactly one QP from the stored iterate: the same step as the oracle's RTI mode (oracle/cpu
    ORACLE_RTI) from the same converged iterate and the same perturbed initial state, to 1e-6."""
    from mpc4rl_amd import MPCBatch, cartpole_ocp, chain_mass_ocp, linear_system_ocp
    from oracle.problems import make_cartpole, make_chain_mass, make_linear_system
    rng = np.random.default_rng(3)
    if name == "cartpole":
        ocp, P, B = cartpole_ocp(), make_cartpole(), 48
        x0 = rng.uniform(-1, 1, (B, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
        x0[: B // 2] = 0.0
        x0[: B // 2, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B // 2)
        x1 = x0 + rng.normal(0, 1, (B, 4)) * np.array([0.01, 0.05, 0.02, 0.05])
    elif name == "linear":
        ocp, P, B = linear_system_ocp(discount_factor=0.99), make_linear_system(gamma=0.99), 16
        x0 = np.column_stack([rng.uniform(0.2, 0.8, B), rng.uniform(-0.4, 0.4, B)])
        x1 = x0 + rng.normal(0, 0.02, (B, 2))
    else:
        ocp, P, B = chain_mass_ocp(), make_chain_mass(), 6
        x0 = np.tile(ocp.x0, (B, 1))
        x0[:, 12:] += rng.normal(0, 1e-2, (B, 9))
        x1 = x0.copy()
        x1[:, 9:12] += rng.normal(0, 5e-3, (B, 3))
        x1[:, 12:] += rng.normal(0, 5e-3, (B, 9))
    mpc = MPCBatch(ocp, B)
    r0 = mpc.solve(x0, cold=True)
    full = oracle_port.solve(P, x0)
    assert bool((r0.status == 0).all()) and np.all(full.status == 0)
    r = mpc.solve(x1, rti=True, sens_v=True)
    ref = oracle_port.solve(P, x1, warm=full, rti=True)
    it = r.iters.cpu().numpy()
    assert np.all(it[:, 0] == 1) and np.all(ref.sqp_iter == 1)
    assert np.array_equal(r.status.cpu().numpy(), ref.status)
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - b) / np.maximum(np.abs(b), 1.0)))
    x, u, pi, _, _ = [t.cpu().numpy() for t in mpc.get_iterate()]
    assert rel(r.u0.cpu().numpy(), ref.u0) < 1e-6 and rel(r.V.cpu().numpy(), ref.V) < 1e-6
    assert rel(x, ref.X) < 1e-6 and rel(u, ref.U) < 1e-6 and rel(pi, ref.PI) < 1e-6
    assert rel(r.dV_dp.cpu().numpy(), ref.dV) < 1e-6
    assert np.abs(it[:, 1] - ref.ipm_iter).max() <= 1


def test_device_envs_vs_reference_formulas():
    """The batched envs on the GPU against the reference's step written out in numpy: cartpole swing-up (Euler, tau = 0.02,
    continuous_cartpole/environment.py:105-134,178-194) and the linear system (linear_system/environment.py:28-58)."""
    import math
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedLinearSystemEnv
    B = 4096
    env = BatchedCartPoleSwingUpEnv(B, device="cuda", seed=3)
    obs = env.reset()
    s = obs.cpu().numpy().copy()
    assert np.all(s[:, [0, 1, 3]] == 0.0) and s[:, 2].min() >= 0.9 * math.pi and s[:, 2].max() <= 1.1 * math.pi
    rng = np.random.default_rng(0)
    g, mc, mp_, l, fm, tau = 9.8, 1.0, 0.1, 0.5, 30.0, 0.02
    for _ in range(25):
        a = rng.uniform(-1, 1, (B, 1))
        obs, rew, term, trunc = env.step(torch.as_tensor(a, device="cuda"))
        x, xd, th, thd = s.T
        force = fm * a[:, 0]
        c, sn = np.cos(th), np.sin(th)
        temp = (force + mp_ * l * thd ** 2 * sn) / (mp_ + mc)
        thacc = (g * sn - c * temp) / (l * (4.0 / 3.0 - mp_ * c ** 2 / (mp_ + mc)))
        xacc = temp - mp_ * l * thacc * c / (mp_ + mc)
        s = np.stack([x + tau * xd, xd + tau * xacc, th + tau * thd, thd + tau * thacc], 1)     # Euler, environment.py:123-127
        assert np.allclose(obs.cpu().numpy(), s, rtol=1e-12, atol=1e-12)
        assert np.allclose(rew.cpu().numpy(), s[:, 0] ** 2 + s[:, 2] ** 2, rtol=1e-12)
        t_ref = (np.abs(s[:, 0]) < 0.1) & (np.abs(s[:, 1]) < 0.1) & (np.abs(s[:, 2]) < math.radians(2.0)) & (np.abs(s[:, 3]) < 0.1)
        assert np.array_equal(term.cpu().numpy(), t_ref) and not bool(trunc.any())
    # the device path is the library's K4 kernels (mpcrl_env_cartpole_step / _reset), not torch ops
    assert env._native()
    # truncation by step count, and the masked reset without a host synchronisation
    short = BatchedCartPoleSwingUpEnv(B, device="cuda", seed=5, max_episode_steps=2)
    short.reset()
    z = torch.zeros(B, 1, device="cuda", dtype=torch.float64)
    _, _, _, tr1 = short.step(z)
    _, _, _, tr2 = short.step(z)
    assert not bool(tr1.any()) and bool(tr2.all()) and tr2.dtype == torch.bool
    mask = torch.zeros(B, dtype=torch.bool, device="cuda")
    mask[::3] = True
    before, steps_before = short.state.clone(), short.steps.clone()
    o = short.reset_where(mask)
    assert torch.equal(o, short.state) and torch.equal(o[~mask], before[~mask]) and torch.equal(short.steps[~mask], steps_before[~mask])
    om = o[mask].cpu().numpy()
    assert np.all(om[:, [0, 1, 3]] == 0.0) and om[:, 2].min() >= 0.9 * math.pi and om[:, 2].max() < 1.1 * math.pi and len(np.unique(om[:, 2])) > 1000
    assert bool((short.steps[mask] == 0).all())
    lin = BatchedLinearSystemEnv(B, device="cuda", lb_noise=-0.1, ub_noise=0.0, seed=1)      # steps through mpcrl_env_linear_step
    o = lin.reset()
    assert bool((o == torch.tensor([0.5, 0.5], device="cuda", dtype=o.dtype)).all())
    A, Bm = np.array([[0.9, 0.35], [0.0, 1.1]]), np.array([[0.0813], [0.2]])
    prev = o.cpu().numpy()
    a = rng.uniform(-1, 1, (B, 1))
    o2, cost, _, _ = lin.step(torch.as_tensor(a, device="cuda"))
    o2 = o2.cpu().numpy()
    noise = o2 - prev @ A.T - a @ Bm.T
    assert np.all(noise[:, 0] >= -0.1 - 1e-12) and np.all(noise[:, 0] <= 1e-12) and np.abs(noise[:, 1]).max() < 1e-12
    viol_lo = ((np.array([0.0, -1.0]) - o2) > 0).any(1) * 1e2
    viol_hi = ((o2 - np.array([1.0, 1.0])) > 0).any(1) * 1e2
    assert np.allclose(cost.cpu().numpy(), 0.5 * (o2 ** 2).sum(1) + 0.5 * (a ** 2).sum(1) + viol_lo + viol_hi, rtol=1e-12)


def test_td3_closed_loop_on_gpu(oracle_port):
    """BASELINE config 5, one rank: 4096 batched cartpole environments -> MPC actor (one launch per step, warm-started, cold only
    where an episode ended) -> replay -> critic TD update -> deterministic policy gradient through du0*/dtheta.  Checks the hot-path
    pieces against the oracle on recorded samples and the loop's invariants."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    from oracle.problems import make_cartpole
    E = 4096
    ocp = cartpole_ocp()
    env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=0, max_episode_steps=3)      # every environment is truncated at step 3
    agent = BatchedTD3(ocp, env, batch_size=512, buffer_steps=8, policy_delay=2, lr_actor=1e-6, seed=1)
    # the swing-up start saturates the input (u0* = -30, du0*/dtheta = 0): put the upper half of the environments near the upright
    # position so that the policy gradient has something to work with
    g = torch.Generator(device="cuda").manual_seed(7)
    near = (torch.rand(E // 2, 4, generator=g, device="cuda", dtype=env.state.dtype) * 2 - 1) * torch.tensor([0.5, 1.0, 0.3, 1.0], device="cuda", dtype=env.state.dtype)
    env.state[E // 2:] = near
    agent.obs = env.state.clone()
    obs0 = agent.obs.clone()
    theta0 = agent.theta.clone()
    st = agent.collect(5)
    assert st["converged_fraction"] == 1.0 and E <= st["episodes_ended"] <= E + 8 and agent.buffer.size() == 5 * E   # (+ the odd terminated one)
    # step 0 of the roll-out: the stored action is the MPC policy (scaled, plus exploration noise of sigma 0.1, clipped)
    ref = oracle_port.solve(make_cartpole(), obs0[:64].double().cpu().numpy(), flags=0)
    a_ref = 2.0 * (ref.u0 + 30.0) / 60.0 - 1.0
    a0 = agent.buffer.act[0, :64].cpu().numpy()
    dev = a0 - np.clip(a_ref, -1, 1)          # N(0, 0.1) exploration noise, clipped into [-1, 1] (the swing-up start saturates at -1)
    assert np.abs(dev).max() < 0.6 and 0.0 < np.abs(dev).mean() < 0.15 and a0.min() >= -1.0 and a0.max() <= 1.0
    # the environments that ended were re-drawn from the reset distribution and their next solve started cold (it converged)
    o3 = agent.buffer.obs[3]
    fresh = (o3[:, [0, 1, 3]].abs().max(1).values == 0.0) & (o3[:, 2] >= 0.9 * np.pi - 1e-6) & (o3[:, 2] <= 1.1 * np.pi + 1e-6)
    assert int(fresh.sum()) >= E - 8
    tr = agent.train(4)
    assert np.isfinite(tr["critic_loss"]) and agent.n_updates == 4
    step = (agent.theta - theta0).cpu().numpy()
    assert np.abs(step[:3]).max() > 0.0 and np.all(step[3:] == 0.0) and np.all(np.isfinite(step))
    assert torch.equal(agent.actor.mpc.get_theta(), agent.theta) and torch.equal(agent.target_mpc.mpc.get_theta(), agent.theta_target)
    # the policy-gradient ingredients on a recorded batch against the oracle: u0*(s_i) and du0*/dtheta_i
    obs = agent.buffer.obs[1, :48].double()
    rp = agent.pi_mpc.mpc
    from mpc4rl_amd import MPCBatch
    m = MPCBatch(ocp, 48)
    m.set_theta(theta0)
    r = m.solve(obs, sens_pi=True, cold=True)
    o = oracle_port.solve(make_cartpole(), obs.cpu().numpy(), p=np.tile(theta0.cpu().numpy(), (48, 1)))
    assert np.allclose(r.u0.cpu().numpy(), o.u0, rtol=1e-6, atol=1e-8) and np.allclose(r.dpi_dp.cpu().numpy(), o.dpi, rtol=1e-5, atol=1e-7)


def test_per_instance_cold_mask(oracle_port):
    """mpcrl_set_cold_mask: the masked instances of a warm call restart from the cold iterate (bitwise what a cold call gives them),
    the others continue from their stored iterate."""
    from mpc4rl_amd import MPCBatch, cartpole_ocp
    B = 90
    rng = np.random.default_rng(2)
    x0 = np.zeros((B, 4))
    x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
    x1 = x0 + rng.normal(0, 0.01, (B, 4))
    mask = torch.as_tensor(rng.uniform(size=B) < 0.4, device="cuda")
    a = MPCBatch(cartpole_ocp(), B)
    a.solve(x0, cold=True, reorder=False)
    ra = a.solve(x1, cold_mask=mask, reorder=False)
    cold = MPCBatch(cartpole_ocp(), B).solve(x1, cold=True, reorder=False)
    warm = MPCBatch(cartpole_ocp(), B)
    warm.solve(x0, cold=True, reorder=False)
    rw = warm.solve(x1, reorder=False)
    m = mask.cpu().numpy()
    assert torch.equal(ra.u0[mask], cold.u0[mask]) and torch.equal(ra.iters[mask], cold.iters[mask]) and torch.equal(ra.V[mask], cold.V[mask])
    assert torch.equal(ra.u0[~mask], rw.u0[~mask]) and torch.equal(ra.iters[~mask], rw.iters[~mask])
    assert int(ra.iters[~mask][:, 0].max()) < int(ra.iters[mask][:, 0].min())        # warm starts need far fewer SQP iterations
    rb = a.solve(x1, reorder=False)                                                   # the mask was one-shot
    assert int(rb.iters[:, 0].max()) == 0


def test_td3_graph_replay_mode():
    """BatchedTD3.enable_graphs: the roll-out step and the two halves of an update captured into HIP graphs (the library calls are
    capture-safe) and replayed.  The replays must do what the eager calls do: the replay buffer fills and wraps, the environments
    move and reset, the critic changes at every update, theta and the Polyak target move at every policy_delay-th one, the solves of
    the closed loop converge, and the roll-out statistics agree with the buffer's contents."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    E = 512
    env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=3, max_episode_steps=12)      # short episodes: resets happen inside replays
    agent = BatchedTD3(cartpole_ocp(), env, batch_size=E, buffer_steps=6, policy_delay=2, lr_actor=1e-3, seed=2)
    agent.collect(2)
    agent.enable_graphs()
    assert agent.buffer.full and agent._graphs is not None
    crit = lambda: torch.cat([p.detach().reshape(-1) for p in agent.critic.parameters()]).clone()
    c_prev, th_prev, tt0, pos0, n0 = crit(), agent.theta.clone(), agent.theta_target.clone(), agent.buffer.pos, agent.n_updates
    obs_prev = agent.obs.clone()
    moved_theta = 0
    agent.collect(0)
    for i in range(8):
        agent.collect(1, stats=False)
        agent.train(1, stats=False)
        torch.cuda.synchronize()
        assert int(agent.buffer.pos_t.item()) == agent.buffer.pos == (pos0 + i + 1) % 6          # device and host positions in step
        c = crit()
        assert bool(torch.isfinite(c).all()) and float((c - c_prev).abs().max()) > 0.0               # every update moves the critic
        assert not torch.equal(agent.obs, obs_prev)
        moved_theta += int(float((agent.theta - th_prev).abs().max()) > 0.0)
        c_prev, th_prev, obs_prev = c, agent.theta.clone(), agent.obs.clone()
    st = agent.last_stats()
    assert agent.n_updates == n0 + 8 and st["converged_fraction"] > 0.95 and st["episodes_ended"] >= E    # 12-step episodes: everyone ended once
    assert bool(torch.isfinite(agent.theta).all()) and float((agent.theta_target - tt0).abs().max()) > 0.0
    assert 1 <= moved_theta <= 4                                                                          # policy_delay = 2: at most every other update
    # the handles hold the parameters the agent holds (set_theta replays as a captured device-to-device copy of the in-place tensor)
    assert torch.equal(agent.actor.mpc.get_theta(), agent.theta) and torch.equal(agent.target_mpc.mpc.get_theta(), agent.theta_target)
    # the newest replay row is the transition the last replayed step produced: its next_obs, un-reset, continues its obs
    last = (agent.buffer.pos - 1) % 6
    assert bool(torch.isfinite(agent.buffer.next_obs[last]).all()) and float(agent.buffer.act[last].abs().max()) <= 1.0


def test_float32_observations_through_the_native_env_kernels():
    """dtype=torch.float32 (what the reference's gymnasium envs return, continuous_cartpole/environment.py:166,186): the state stays
    fp64 (the reference's numpy state), the observation is stored as float by the SAME library kernels — one device path."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedLinearSystemEnv
    B = 1024
    e32, e64 = (BatchedCartPoleSwingUpEnv(B, device="cuda", seed=4, dtype=dt) for dt in (torch.float32, torch.float64))
    assert e32._native() and e32.state.dtype == torch.float64
    o32, o64 = e32.reset(), e64.reset()
    assert o32.dtype == torch.float32 and torch.equal(o32, o64.float()) and torch.equal(e32.state, e64.state)
    rng = np.random.default_rng(1)
    for _ in range(5):
        a = torch.as_tensor(rng.uniform(-1, 1, (B, 1)), device="cuda")
        sp = e32.state.data_ptr()
        o32, r32, t32, _ = e32.step(a.float())
        o64, r64, t64, _ = e64.step(a.float())
        assert e32.state.data_ptr() == sp                                       # updated in place: a captured step keeps working
        assert o32.dtype == torch.float32 and torch.equal(o32, o64.float()) and torch.equal(e32.state, e64.state)
        assert torch.equal(r32, r64) and torch.equal(t32, t64)
    mask = torch.arange(B, device="cuda") % 2 == 0
    assert torch.equal(e32.reset_where(mask), e64.reset_where(mask).float())
    l32 = BatchedLinearSystemEnv(B, device="cuda", seed=2, dtype=torch.float32)
    l64 = BatchedLinearSystemEnv(B, device="cuda", seed=2)
    l32.reset(), l64.reset()
    a = torch.as_tensor(rng.uniform(-1, 1, (B, 1)), device="cuda")
    o32, c32, _, _ = l32.step(a)
    o64, c64, _, _ = l64.step(a)
    assert o32.dtype == torch.float32 and torch.equal(o32, o64.float()) and torch.equal(c32, c64)


def test_handleless_kernels_refuse_host_memory():
    """mpcrl_weighted_grad_sum / mpcrl_env_* carry no handle: they launch on the device that owns the memory they are given and
    return MPCRL_E_ARG for anything that is not device memory (instead of launching on whatever device happens to be current)."""
    import ctypes as C
    from mpc4rl_amd import _lib
    lib = _lib.load()
    host = (C.c_double * 16)()
    dev = torch.zeros(16, dtype=torch.float64, device="cuda")
    assert lib.mpcrl_weighted_grad_sum(C.c_void_p(dev.data_ptr()), 4, None, 4, 2, C.cast(host, C.c_void_p), None) == -1
    assert lib.mpcrl_weighted_grad_sum(C.c_void_p(dev.data_ptr()), 4, None, 4, 2, C.c_void_p(dev.data_ptr() + 64), None) == 0
    par = (C.c_double * 9)(9.8, 1.0, 0.1, 0.5, 30.0, 0.02, 0.1, 0.03, 500.0)
    assert lib.mpcrl_env_cartpole_step(par, 2, C.cast(host, C.c_void_p), C.c_void_p(dev.data_ptr()), C.c_void_p(dev.data_ptr()), None, 0,
                                       C.c_void_p(dev.data_ptr()), C.c_void_p(dev.data_ptr()), C.c_void_p(dev.data_ptr()), None) == -1


def test_enable_graphs_leaves_the_training_trajectory_alone():
    """enable_graphs warms the capture stream up with two real updates and one real roll-out step; everything they touched is put
    back (critics, optimiser state, theta / theta', generators, environments, replay buffer, warm-start iterates), so that an agent
    that enables graphs continues exactly where it was — and then follows the eager agent's trajectory (same kernels, same random
    numbers: the generators are registered with the graphs)."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    E = 256

    def make():
        env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=11, max_episode_steps=9)
        ag = BatchedTD3(cartpole_ocp(), env, batch_size=E, buffer_steps=4, policy_delay=2, lr_actor=1e-3, seed=5)
        ag.collect(4)                       # the buffer is full: enable_graphs collects nothing on its own
        ag.train(1)                         # n_updates = 1: the next update is a policy update
        return ag

    flat = lambda ag: torch.cat([p.detach().reshape(-1).double() for p in ag.critic.parameters()] + [ag.theta, ag.theta_target,
                                 ag.env.state.reshape(-1), ag.obs.reshape(-1).double(), ag.buffer.obs.reshape(-1).double()])
    a, b = make(), make()
    assert torch.equal(flat(a), flat(b))
    before, pos, nup = flat(b).clone(), (b.buffer.pos, b.buffer.full), b.n_updates
    b.enable_graphs()
    torch.cuda.synchronize()
    assert torch.equal(flat(b), before) and (b.buffer.pos, b.buffer.full) == pos and b.n_updates == nup
    assert int(b.buffer.pos_t.item()) == b.buffer.pos
    for _ in range(3):
        for ag in (a, b):
            ag.collect(1, stats=False)
            ag.train(1, stats=False)
    torch.cuda.synchronize()
    fa, fb = flat(a), flat(b)
    assert bool(torch.isfinite(fb).all())
    assert float((fa - fb).abs().max() / fa.abs().max()) < 1e-6, float((fa - fb).abs().max())
    with pytest.raises(RuntimeError):       # a CPU environment cannot be replayed (its state would be frozen at the captured address)
        BatchedTD3(cartpole_ocp(), BatchedCartPoleSwingUpEnv(E, device="cpu"), batch_size=E, buffer_steps=2, device="cuda").enable_graphs()


def test_td3_config5_full_size_on_one_gpu():
    """BASELINE config 5 at its full size — 32 768 cartpole environments — on ONE GPU (the 8-GPU job gives each rank 4096 of them):
    two closed-loop steps + two TD3 updates on batches of 32 768 through the `nccl` (= RCCL) process group at world size 1, ONE
    collective per update (scripts/cartpole_mpc_as_td3_agent_closed_loop.py:40-67 is the loop this batches)."""
    import socket
    import torch.distributed as dist
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    E = 32768
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    calls = []
    real = dist.all_reduce
    try:
        env = BatchedCartPoleSwingUpEnv(E, device=dev, seed=0)
        agent = BatchedTD3(cartpole_ocp(), env, batch_size=E, buffer_steps=4, policy_delay=1, lr_actor=1e-6, seed=0, device=dev)
        agent._world = lambda: 2            # take the collective's path although the group has one rank
        agent._allreduce = lambda flat: (calls.append(int(flat.numel())), real(flat, group=agent.group), flat.mul_(2.0))[-1]
        th0 = agent.theta.clone()
        st = agent.collect(2)
        assert agent.buffer.size() == 2 * E and st["converged_fraction"] > 0.999 and np.isfinite(st["mean_reward"])
        for _ in range(2):
            tr = agent.train(1)
        torch.cuda.synchronize()
        n_crit = sum(p.numel() for p in agent.critic.parameters())
        assert calls == [n_crit + agent.theta.numel() + 1] * 2                  # ONE message per update: critic + theta gradients + count
        assert np.isfinite(tr["critic_loss"]) and bool(torch.isfinite(agent.theta).all()) and bool(torch.isfinite(agent.theta_target).all())
        assert torch.equal(agent.theta[3:], th0[3:])                            # only the model block is learned
        r = agent.pi_mpc.mpc.solve(agent.buffer.obs[0].double(), sens_pi=True, cold=True)
        assert float((r.status == 0).double().mean()) > 0.999 and bool(torch.isfinite(r.dpi_dp).all())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sb3_shaped_td3_policy():
    """`MPCTD3Policy` keeps the attribute and method names an SB3 TD3 algorithm uses on rlmpc/td3/policies.py:225-361 (stable-baselines3
    itself is not installable here): actor / actor_target / critic / critic_target, make_actor / make_critic, forward / _predict,
    set_training_mode; forward(obs) is one batched solve and equals the scaled u0* of the MPC; the target critic starts as a copy."""
    from types import SimpleNamespace
    from mpc4rl_amd import MPCBatch, MPCTD3Policy, cartpole_ocp
    ocp = cartpole_ocp()
    B = 48
    pol = MPCTD3Policy(SimpleNamespace(shape=(4,)), SimpleNamespace(shape=(1,)), lambda progress: 1e-3, ocp, net_arch=[32, 32], n_critics=2,
                       batch=B, device="cuda")
    for name in ("actor", "actor_target", "critic", "critic_target"):
        assert hasattr(pol, name)
    assert isinstance(pol.critic.optimizer, torch.optim.Adam) and pol.actor.lr == 1e-3
    rng = np.random.default_rng(3)
    obs = torch.as_tensor(rng.uniform(-1, 1, (B, 4)) * np.array([0.5, 1.0, 0.3, 1.0]), dtype=torch.float32, device="cuda")
    a = pol(obs)
    assert a.shape == (B, 1) and a.dtype == torch.float32 and float(a.abs().max()) <= 1.0 + 1e-6
    ref = MPCBatch(ocp, B).solve(obs.to(torch.float64), cold=True)
    assert torch.allclose(a.to(torch.float64), ref.u0 / 30.0, atol=1e-6)         # scale_action with u in [-30, 30] (mpc.py:290-301)
    assert torch.equal(pol._predict(obs), pol.forward(obs))                      # warm call from the stored iterate: the same action
    q, qt = pol.critic(obs, a), pol.critic_target(obs, a)
    assert len(q) == 2 and all(torch.equal(x, y) for x, y in zip(q, qt))
    pol.set_training_mode(False)
    assert not pol.critic.training and not pol.critic_target.training
    assert pol.actor.parameters().shape == (ocp.n_p,)


@pytest.mark.gpu
def test_iterate_rows_moves():
    """mpcrl_get_iterate_rows / mpcrl_set_iterate_rows (ABI 120): whole iterates between a handle and caller-owned tables by row
    index — what get_iterate + index_copy / index + set_iterate do in four passes, in one launch each."""
    from mpc4rl_amd import MPCBatch, cartpole_ocp
    B, R = 96, 300
    rng = np.random.default_rng(11)
    x0 = np.zeros((B, 4))
    x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
    a = MPCBatch(cartpole_ocp(), B)
    ra = a.solve(x0, cold=True)
    x, u, pi, bnd, _ = a.get_iterate()
    lens = (21 * 4, 20 * 1, 20 * 4, 10 * 21 * 5)
    tabs = [torch.full((R, n), -7.0, dtype=torch.float64, device="cuda") for n in lens]
    rows = torch.as_tensor(rng.permutation(R)[:B], device="cuda")
    a.get_iterate_rows(*tabs, index=rows)
    for t, src in zip(tabs, (x, u, pi, bnd)):
        assert torch.equal(t[rows], src.reshape(B, -1))
        untouched = torch.ones(R, dtype=torch.bool, device="cuda")
        untouched[rows] = False
        assert bool((t[untouched] == -7.0).all())
    # a second handle started from those rows (in another order) is converged at once: zero SQP iterations, same answer
    order = torch.as_tensor(rng.permutation(B), device="cuda")
    b = MPCBatch(cartpole_ocp(), B)
    b.set_iterate_rows(*tabs, index=rows[order].contiguous())
    rb = b.solve(torch.as_tensor(x0, device="cuda")[order])
    assert int(rb.iters[:, 0].max()) == 0 and torch.equal(rb.u0, ra.u0[order]) and b.duals_valid
    xb = b.get_iterate()[0]
    assert torch.equal(xb, x[order])
    # identity index, primal part only: next solve starts its interior point from the default point (as set_iterate(bnd=None))
    c, d = MPCBatch(cartpole_ocp(), B), MPCBatch(cartpole_ocp(), B)
    c.set_iterate_rows(x.reshape(B, -1), u.reshape(B, -1), pi.reshape(B, -1), None)
    d.set_iterate(x, u, pi, None)
    assert not c.duals_valid and not d.duals_valid
    rc, rd = c.solve(x0), d.solve(x0)
    assert torch.equal(rc.u0, rd.u0) and torch.equal(rc.iters, rd.iters)
    # a negative row skips the instance: set leaves its stored iterate alone, get records nothing for it
    keep = torch.arange(B, device="cuda") % 2 == 0
    e = MPCBatch(cartpole_ocp(), B)
    e.solve(x0 * 0.99, cold=True)
    xe = e.get_iterate()[0].clone()
    e.set_iterate_rows(*tabs, index=torch.where(keep, -1, rows))
    xe2 = e.get_iterate()[0]
    assert torch.equal(xe2[keep], xe[keep]) and torch.equal(xe2[~keep], x[~keep])
    tabs2 = [torch.full((R, n), -7.0, dtype=torch.float64, device="cuda") for n in lens]
    a.get_iterate_rows(*tabs2, index=torch.where(keep, -1, rows))
    assert bool((tabs2[0][rows[keep]] == -7.0).all()) and torch.equal(tabs2[0][rows[~keep]], x.reshape(B, -1)[~keep])
    with pytest.raises(ValueError):
        a.get_iterate_rows(*tabs, index=rows.to(torch.int32))


@pytest.mark.gpu
def test_td3_replay_iterates():
    """BatchedTD3(replay_iterates=True): every transition keeps the iterate the roll-out policy's solve ended with, and the two replay
    solves of an update start from it instead of from the cold iterate (round 6).  (a) The stored rows ARE the roll-out solutions:
    a fresh handle started from them at the stored observations is converged at once with the stored action.  (b) The target actor's
    and the policy's answers on a sampled batch equal the cold solves' at 1e-6 (same KKT point, far fewer iterations).  (c) A few
    closed-loop steps in eager and in graph mode: converged, finite, critic and theta move."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, MPCBatch, cartpole_ocp
    E = 512
    ocp = cartpole_ocp()
    env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=5, max_episode_steps=9)       # short episodes: done rows and resets in the buffer
    agent = BatchedTD3(ocp, env, batch_size=E, buffer_steps=8, policy_delay=2, lr_actor=1e-4, seed=1, replay_iterates=True, action_noise=0.0)
    agent.collect(8)
    buf = agent.buffer
    assert buf.full and bool(buf.iter_ok.float().mean() > 0.95)
    # (a) slot 5: observations and their stored iterates
    s = 5
    rows = s * E + torch.arange(E, device="cuda")
    m = MPCBatch(ocp, E)
    m.set_iterate_rows(*buf.iters, index=rows)
    r = m.solve(buf.obs[s].double())
    ok = buf.iter_ok[s]
    assert int(r.iters[ok][:, 0].max()) == 0
    a_stored = buf.act[s][ok].double()                                            # action_noise = 0: the stored action is the policy's
    assert float((agent.actor.scale_action(r.u0[ok]) - a_stored).abs().max()) < 1e-6
    # (b) a sampled batch, warm from the stored iterates against cold
    obs, nxt, act, rew, done = buf.sample(E, agent.gen)
    i_s, ok_s, i_n, ok_n = buf.iterates_of_last_sample()
    step, nstep = buf.last_idx // E, (buf.last_idx // E + 1) % buf.cap
    cont = (done == 0) & (nstep != buf.pos)
    assert torch.equal(i_n, torch.where(cont, nstep * E + buf.last_idx % E, buf.last_idx)) and bool(cont.any()) and bool((~cont).any())
    for handle, states, rows_, okr in ((agent.target_mpc.mpc, nxt, i_n, ok_n), (agent.pi_mpc.mpc, obs, i_s, ok_s)):
        states = states.double()
        handle.set_iterate_rows(*buf.iters, index=rows_.contiguous())
        rw = handle.solve(states, sens_pi=True, cold_mask=~okr)
        rc = MPCBatch(ocp, E).solve(states, sens_pi=True, cold=True)
        both = (rw.status == 0) & (rc.status == 0)
        assert float(both.float().mean()) > 0.97
        err = ((rw.u0 - rc.u0).abs() / rc.u0.abs().clamp(min=1.0))[both]
        same = err[:, 0] < 1e-6                                                   # (a swing-up state can have two local minima: count, do not hide)
        print("replay solve warm vs cold: same KKT point on %.4f of the batch, SQP iterations %.2f vs %.2f (max %d vs %d)" % (
            float(same.float().mean()), float(rw.iters[:, 0].double().mean()), float(rc.iters[:, 0].double().mean()), int(rw.iters[:, 0].max()), int(rc.iters[:, 0].max())))
        assert float(same.float().mean()) > 0.99 and float(rw.iters[:, 0].double().mean()) < 0.5 * float(rc.iters[:, 0].double().mean())
        dw, dc = torch.nan_to_num(rw.dpi_dp[both][same][:, 0, :3]), torch.nan_to_num(rc.dpi_dp[both][same][:, 0, :3])
        assert float(((dw - dc).abs() / dc.abs().amax(1, keepdim=True).clamp(min=1.0)).max()) < 1e-4
    # (c) eager steps, then graph mode
    th0 = agent.theta.clone()
    crit = lambda: torch.cat([p.detach().reshape(-1) for p in agent.critic.parameters()]).clone()
    c0 = crit()
    agent.collect(2)
    tr = agent.train(2)
    assert np.isfinite(tr["critic_loss"]) and float((crit() - c0).abs().max()) > 0.0 and float((agent.theta - th0).abs().max()) > 0.0
    agent.enable_graphs()
    agent.collect(0)
    for _ in range(6):
        agent.collect(1, stats=False)
        agent.train(1, stats=False)
    torch.cuda.synchronize()
    st = agent.last_stats()
    assert st["converged_fraction"] > 0.95 and bool(torch.isfinite(agent.theta).all()) and bool(torch.isfinite(crit()).all())
    assert int(buf.pos_t.item()) == buf.pos


@pytest.mark.gpu
def test_policy_action_kernel_and_fused_td3_glue():
    """mpcrl_policy_action (round 6): the actor's output stage — scale_action (mpc.py:290-301), exploration / target-policy noise, clips,
    the failed-solve mask — in one launch.  (a) Bit for bit the torch expressions it replaces, failed and non-finite rows included.
    (b) A TD3 loop run with it, with mpcrl_replay_sample and with mpcrl_td3_cartpole_collect (BatchedTD3._fused / _fused_sample /
    _fused_collect, the defaults on the GPU) reproduces the loop on the framework's own launches bit for bit: replay table, iterate
    tables and flags, environment state, critic losses, theta, critics; the statistics to rounding (another summation order)."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, MPCBatch, cartpole_ocp
    from mpc4rl_amd.td3 import MPCActor
    ocp = cartpole_ocp()
    B = 300
    rng = np.random.default_rng(8)
    x0 = rng.uniform(-1, 1, (B, 4)) * np.array([2.0, 3.0, np.pi, 5.0])        # the whole box: some solves end with status 2 / 4
    actor = MPCActor(ocp, B)
    actor.mpc.set_options(max_iter=12)
    r = actor.mpc.solve(x0, cold=True)
    r.u0[5] = float("nan")
    r.u0[6] = float("inf")
    st = r.status
    assert int((st == 0).sum()) > 0 and int((st != 0).sum()) > 0
    eps = torch.randn(B, 1, device="cuda")
    for sigma, clip, acc2 in ((0.1, 0.0, True), (0.2, 0.5, False)):
        a, ok = actor.action(r, eps, sigma=sigma, noise_clip=clip, accept_status2=acc2)
        ok_t = torch.isfinite(r.u0).all(dim=1) & ((st == 0) | (st == 2) if acc2 else (st == 0))
        n = sigma * eps
        if clip > 0:
            n = n.clamp(-clip, clip)
        a_t = (torch.where(ok_t[:, None], actor.scale_action(torch.nan_to_num(r.u0)), torch.zeros_like(r.u0)).to(torch.float32) + n).clamp(-1.0, 1.0)
        assert torch.equal(ok, ok_t) and torch.equal(a, a_t)
    a0, ok0 = actor.action(r)                                              # no noise: the scaled action itself, unclipped
    assert torch.equal(a0, torch.where(ok0[:, None], actor.scale_action(torch.nan_to_num(r.u0)), torch.zeros_like(r.u0)).to(torch.float32))
    with pytest.raises(ValueError):
        actor.action(r, eps.double())
    # (b)
    res = {}
    for fused in (False, True):
        env = BatchedCartPoleSwingUpEnv(256, device="cuda", seed=0, max_episode_steps=7)
        ag = BatchedTD3(ocp, env, batch_size=256, buffer_steps=6, policy_delay=2, lr_actor=1e-4, seed=0, replay_iterates=True, fused_critic=False)
        assert ag._fused and ag._fused_sample and ag._fused_collect and not ag._fused_critic
        ag._fused = ag._fused_sample = ag._fused_collect = fused
        ag.collect(6)
        losses = []
        for _ in range(6):
            ag.collect(1)
            losses.append(ag.train(1)["critic_loss"])
        st = ag.last_stats()
        res[fused] = (ag.buffer.data.clone(), losses, ag.theta.clone(), torch.cat([p.detach().reshape(-1) for p in ag.critic.parameters()]),
                      [t.clone() for t in ag.buffer.iters] + [ag.buffer.iter_ok.clone(), env.state.clone(), env.steps.clone(), ag.obs.clone().double(),
                                                              ag.buffer.pos_t.clone()], st, (ag.buffer.pos, ag.buffer.full))
    assert torch.equal(res[False][0], res[True][0]) and res[False][1] == res[True][1]
    assert torch.equal(res[False][2], res[True][2]) and torch.equal(res[False][3], res[True][3])
    assert all(torch.equal(x, y) for x, y in zip(res[False][4], res[True][4])) and res[False][6] == res[True][6]
    sa, sb = res[False][5], res[True][5]
    assert abs(sa["mean_reward"] - sb["mean_reward"]) < 1e-12 * abs(sa["mean_reward"]) and sa["converged_fraction"] == sb["converged_fraction"]
    assert sa["episodes_ended"] == sb["episodes_ended"]


def _critic_reference(critic, target, rows, nx, nu, a_next, ok_u, gamma, dtype):
    """the torch expressions of BatchedTD3._update_pre (autograd) in ``dtype``: (loss, flat gradient, ok_b)"""
    import copy
    cr, tg = copy.deepcopy(critic).to(dtype), copy.deepcopy(target).to(dtype)
    row = rows[:, : 2 * nx + nu + 2]
    ok_b = ok_u & torch.isfinite(row).all(dim=1) & torch.isfinite(a_next).all(dim=1)
    safe = torch.where(ok_b[:, None], row, 0.0).to(dtype)
    an = torch.where(ok_b[:, None], a_next, 0.0).to(dtype)
    obs_s, nxt_s, act_s, rew, done = safe[:, :nx], safe[:, nx: 2 * nx], safe[:, 2 * nx: 2 * nx + nu], safe[:, 2 * nx + nu], safe[:, 2 * nx + nu + 1]
    with torch.no_grad():
        qn = tg(nxt_s, an)
        q_next = (torch.min(*qn) if len(qn) > 1 else qn[0]).squeeze(1)
        y = torch.where(ok_b, rew + gamma * (1.0 - done) * q_next, 0.0)
    qs = cr(obs_s, act_s)
    loss = sum((torch.where(ok_b, q.squeeze(1) - y, 0.0) ** 2).sum() for q in qs) / ok_b.to(dtype).sum().clamp(min=1.0)
    loss.backward()
    return loss.detach(), torch.cat([p.grad.reshape(-1) for p in cr.parameters()]), ok_b


@pytest.mark.parametrize("B,nx,nu,n_critics", [(4096, 4, 1, 2), (257, 4, 1, 2), (37, 9, 3, 1), (16, 2, 1, 2), (5, 33, 3, 2)])
def test_critic_td_grad_and_dq_da_vs_autograd(B, nx, nu, n_critics):
    """mpcrl_critic_td_grad / mpcrl_critic_dq_da (round 6, ABI 130): the TD target, the twin-critic loss, its gradient with respect to the
    critics' parameters and dQ_1/da — hand-written forward and backward passes of the [obs | action] -> 64 -> 64 -> 1 MLPs — against torch
    autograd of the same expressions.  The yardstick is the float64 evaluation: the kernels (fp32 FMAs, fp64 reduction over the batch) must
    be as close to it as torch's own fp32 path is (1e-6 of the largest gradient entry, and within 4 x torch's error).  Rows with a non-finite entry, rows whose target
    solve failed and a ragged last workgroup are part of every case; the selected-out rows are reported back (ok)."""
    from mpc4rl_amd.td3 import ContinuousCritic, critic_dq_da, critic_td_grad, flatten_parameters
    torch.manual_seed(B + nx)
    dev = torch.device("cuda")
    critic, target = ContinuousCritic(nx, nu, n_critics=n_critics).to(dev), ContinuousCritic(nx, nu, n_critics=n_critics).to(dev)
    flat, grad = flatten_parameters(critic)
    flat_t, _ = flatten_parameters(target)
    assert all(p.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() >= grad.data_ptr() for p in critic.parameters())
    rows = torch.randn(B, 2 * nx + nu + 2 + 3, device=dev)[:, : 2 * nx + nu + 2]            # a row stride larger than the row
    rows[:, -1] = (torch.rand(B, device=dev) < 0.1).float()
    a_next = torch.randn(B, nu, device=dev).clamp(-1, 1)
    ok_u = torch.rand(B, device=dev) < 0.9
    if B > 4:
        rows[1, 0], rows[2, nx + 1], rows[3, 2 * nx] = float("nan"), float("inf"), float("-inf")
        a_next[4, 0] = float("nan")
    gamma = 0.99
    out = torch.full((flat.numel() + 7,), -1.0, dtype=torch.float64, device=dev)
    loss, ok, ws = critic_td_grad(rows, nx, nu, a_next, ok_u, flat, flat_t, n_critics, gamma, 0.5, out)
    l64, g64, ok64 = _critic_reference(critic, target, rows, nx, nu, a_next, ok_u, gamma, torch.float64)
    l32, g32, _ = _critic_reference(critic, target, rows, nx, nu, a_next, ok_u, gamma, torch.float32)
    assert torch.equal(ok, ok64) and bool((out[flat.numel():] == -1.0).all())
    scale = float(g64.abs().max())
    err_k = float((2.0 * out[: flat.numel()] - g64).abs().max()) / scale
    err_t = float((g32.double() - g64).abs().max()) / scale
    print(f"B {B} nx {nx} nu {nu} critics {n_critics}: gradient against float64 — kernels {err_k:.1e}, torch float32 {err_t:.1e}; loss {abs(float(loss) - float(l64)) / float(l64):.1e}")
    assert err_k < 1e-6 and err_k < 4.0 * err_t and abs(float(loss) - float(l64)) < 1e-6 * float(l64)
    # the same call again: the same bits (fixed-order reduction), the workspace reused
    out2 = torch.zeros_like(out)
    loss2, _, ws2 = critic_td_grad(rows, nx, nu, a_next, ok_u, flat, flat_t, n_critics, gamma, 0.5, out2, ws)
    assert ws2 is ws and torch.equal(out2[: flat.numel()], out[: flat.numel()]) and torch.equal(loss, loss2)
    # dQ_1/da
    act = torch.randn(B, nu, device=dev).clamp(-1, 1)
    dq, okq = critic_dq_da(rows, nx, act, ok_u, flat)
    okr = ok_u & torch.isfinite(rows[:, :nx]).all(dim=1)
    a64 = act.double().requires_grad_(True)
    import copy
    c64 = copy.deepcopy(critic).double()
    (ref,) = torch.autograd.grad(c64.q1_forward(torch.where(okr[:, None], rows[:, :nx], 0.0).double(), a64).sum(), a64)
    ref = torch.where(okr[:, None], ref, 0.0)
    assert torch.equal(okq, okr)
    assert float((dq.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        critic_td_grad(rows.double(), nx, nu, a_next, ok_u, flat, flat_t, n_critics, gamma, 1.0, out)
    # the policy gradient's contraction with dpi/dtheta (mpcrl_dpg_grad): torch's expressions in float64, non-finite sensitivities on rows
    # that are left in and on rows that are left out
    from mpc4rl_amd.td3 import dpg_grad
    n_p = 83
    dpi = torch.randn(B, nu, n_p, dtype=torch.float64, device=dev)
    dpi[0, 0, 3] = float("nan")
    if B > 4:
        dpi[1] = float("nan")
        dpi[4, nu - 1, 7] = float("inf")
    lo, hi = -torch.rand(nu, dtype=torch.float64, device=dev) - 0.5, torch.rand(nu, dtype=torch.float64, device=dev) + 0.5
    for scale in (True, False):
        msg = torch.full((5 + n_p + 1,), -1.0, dtype=torch.float64, device=dev)
        wsd = dpg_grad(dq, okq, dpi, lo, hi, scale, msg[5:])
        chain = 2.0 / (hi - lo) if scale else torch.ones_like(lo)
        g = torch.einsum("bu,bup->bp", torch.where(okq[:, None], dq.double() * chain, 0.0), torch.nan_to_num(dpi)).sum(0)
        assert bool((msg[:5] == -1.0).all()) and float(msg[-1]) == float(okq.sum())
        fin = g.abs() < 1e300
        assert float((msg[5:-1][fin] - g[fin]).abs().max()) <= 1e-12 * max(1.0, float(g[fin].abs().max())) and bool((msg[5:-1][~fin].abs() > 1e300).all())
        msg2 = torch.zeros_like(msg)
        assert dpg_grad(dq, okq, dpi, lo, hi, scale, msg2[5:], wsd) is wsd and torch.equal(msg2[5:], msg[5:])      # counter left at zero, same bits


def test_td3_loop_with_the_critic_kernels():
    """BatchedTD3 with the critic step on mpcrl_critic_td_grad / mpcrl_critic_dq_da (the default on the GPU for 64 x 64 critics) against the
    same loop on autograd: the same replay table bit for bit (the roll-out does not see the critics), critic losses, critics and theta
    equal to fp32 rounding after six updates — eager and graph-replayed."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    ocp = cartpole_ocp()
    res = {}
    for mode in ("autograd", "kernels", "kernels+graphs", "kernels+graphs, step()"):
        env = BatchedCartPoleSwingUpEnv(256, device="cuda", seed=0, max_episode_steps=7)
        ag = BatchedTD3(ocp, env, batch_size=256, buffer_steps=6, policy_delay=2, lr_actor=1e-4, seed=0, replay_iterates=True, fused_critic=mode != "autograd")
        assert ag._fused_critic == (mode != "autograd")
        ag.collect(6)
        if "graphs" in mode:
            ag.enable_graphs()
        losses = []
        for _ in range(6):
            if mode.endswith("step()"):      # one graph replay per closed-loop step (single rank)
                ag.step()
                losses.append(ag.last_critic_loss())
            else:
                ag.collect(1)
                losses.append(ag.train(1)["critic_loss"])
        res[mode] = (ag.buffer.data.clone(), np.array(losses), ag.theta.clone(), torch.cat([p.detach().reshape(-1) for p in ag.critic.parameters()]),
                     torch.cat([p.detach().reshape(-1) for p in ag.critic_target.parameters()]))
    a, k = res["autograd"], res["kernels"]
    assert torch.equal(a[0], k[0])
    assert np.abs(a[1] - k[1]).max() < 1e-4 * np.abs(a[1]).max()
    assert float((a[2] - k[2]).abs().max()) < 1e-9 and float((a[3] - k[3]).abs().max()) < 2e-5 and float((a[4] - k[4]).abs().max()) < 2e-6
    g, h = res["kernels+graphs"], res["kernels+graphs, step()"]      # (enable_graphs fills the replay table on its own: the two graph forms against each other)
    assert np.isfinite(g[1]).all() and bool(torch.isfinite(g[3]).all())
    assert torch.equal(g[0], h[0]) and np.array_equal(g[1], h[1]) and all(torch.equal(x, y) for x, y in zip(g[2:], h[2:]))


def test_td3_cartpole_collect_kernel_vs_the_kernels_it_stands_for():
    """mpcrl_td3_cartpole_collect (round 6): the roll-out side of a TD3 step after the solve in one launch, against the launches it
    stands for on the same inputs — mpcrl_policy_action, mpcrl_env_cartpole_step, the replay row, mpcrl_env_cartpole_reset — bit for
    bit (failed solves, non-finite controls, terminated and truncated episodes included), the statistics and the device-side write
    position over two calls (wrap-around of a two-slot table)."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, _lib
    from mpc4rl_amd.batch import _ptr
    lib, dev, E = _lib.load(), torch.device("cuda:0"), 1000
    torch.manual_seed(0)
    env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=0, max_episode_steps=3)
    env.reset()
    env.state.copy_(torch.randn(E, 4, dtype=torch.float64, device=dev) * torch.tensor([0.5, 1.0, 3.0, 2.0], device=dev, dtype=torch.float64))
    env.state[:50] *= 0.01                                   # some inside the terminal box after the step
    env.steps.copy_(torch.randint(0, 3, (E,), device=dev))
    lo, hi = torch.tensor([-30.0], dtype=torch.float64, device=dev), torch.tensor([30.0], dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    state1, steps1 = env.state.clone(), env.steps.clone()
    obs, ended = env.state.clone(), torch.zeros(E, dtype=torch.int32, device=dev)
    table, pos = torch.zeros(2, E, 11, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    iter_ok, rows = torch.zeros(2, E, dtype=torch.uint8, device=dev), torch.zeros(E, dtype=torch.int64, device=dev)
    stats, ws = torch.zeros(3, dtype=torch.float64, device=dev), torch.zeros(2 + 3 * 4, dtype=torch.float64, device=dev)
    ref_stats = torch.zeros(3, dtype=torch.float64, device=dev)
    for call in range(3):
        u0 = torch.randn(E, 1, dtype=torch.float64, device=dev) * 20
        u0[7], u0[8] = float("nan"), float("inf")
        status = torch.randint(0, 5, (E,), device=dev, dtype=torch.int32)
        eps, u01 = torch.randn(E, 1, device=dev), torch.rand(E, dtype=torch.float64, device=dev)
        u0[:50], eps[:50], status[:50] = 0.0, 0.0, 0                 # (no force: the near-origin states stay in the terminal box)
        # the launches it stands for
        a, ok = torch.empty(E, 1, device=dev), torch.empty(E, dtype=torch.uint8, device=dev)
        assert lib.mpcrl_policy_action(_ptr(u0), _ptr(status), _ptr(eps), _ptr(lo), _ptr(hi), E, 1, 1, 0.1, 0.0, 1, _ptr(a), _ptr(ok), st) == 0
        obs_before = env.state.clone()
        nxt, rew, term, trunc = env.step(a)
        done = term | trunc
        m8, new_obs = done.to(torch.uint8), torch.empty(E, 4, dtype=torch.float64, device=dev)
        assert lib.mpcrl_env_cartpole_reset(E, _ptr(env.state), _ptr(env.steps), _ptr(m8), _ptr(u01), _ptr(new_obs), 0, st) == 0
        ref_stats += torch.stack([rew.sum(), (status == 0).sum().double(), done.sum().double()])
        # the one launch
        rc = lib.mpcrl_td3_cartpole_collect(env._par(), E, _ptr(state1), _ptr(steps1), _ptr(u0), _ptr(status), _ptr(eps), _ptr(u01), -30.0, 30.0, 1, 0.1, _ptr(obs),
                                            _ptr(ended), _ptr(table), 2, -1.0, _ptr(pos), _ptr(iter_ok), _ptr(rows), _ptr(stats), _ptr(ws), st)
        assert rc == 0
        slot = call % 2
        row = torch.cat([obs_before.float(), nxt.float(), a, (-1.0 * rew).float()[:, None], term.float()[:, None]], dim=1)
        assert torch.equal(table[slot], row) and int(pos) == (call + 1) % 2
        assert torch.equal(state1, env.state) and torch.equal(steps1, env.steps) and torch.equal(obs, new_obs) and torch.equal(ended.bool(), done)
        assert torch.equal(iter_ok[slot].bool(), ok.bool() & (status == 0)) and torch.equal(rows, slot * E + torch.arange(E, device=dev))
        assert int(done.sum()) > 0 and (call > 0 or int(term.sum()) > 0)
    assert float((stats - ref_stats).abs().max()) < 1e-9 * float(ref_stats.abs().max()) and torch.equal(stats[1:], ref_stats[1:])
    assert float(ws[0]) == 0.0


def test_replay_sample_excluding_the_slot_being_written():
    """mpcrl_replay_sample with exclude_pos (the pipelined TD3 loop: a roll-out writes slot pos while the batch is drawn): the rows of the
    other cap - 1 slots in the order after pos, the excluded slot never touched, and pos standing in for the write position in the choice
    of the iterate for next_obs — against the index expressions in torch."""
    from mpc4rl_amd.td3 import DeviceReplayBuffer
    dev, E, cap, nx, nu, n = torch.device("cuda"), 37, 5, 4, 1, 4000
    buf = DeviceReplayBuffer(cap, E, nx, nu, dev, iterate_dims=(3, 2, 2, 4))
    buf.data.copy_(torch.randn_like(buf.data))
    buf.data[..., -1] = (torch.rand(cap, E, device=dev) < 0.3).float()
    buf.iter_ok.copy_(torch.rand(cap, E, device=dev) < 0.8)
    buf.pos, buf.full = 2, True
    buf.pos_t.fill_(2)
    gen = torch.Generator(device=dev).manual_seed(3)
    for pos in (0, 3, 4):
        ex = torch.tensor([pos], dtype=torch.int64, device=dev)
        st = gen.get_state()
        obs, nxt, act, rew, done = buf.sample_fused(n, gen, exclude_pos=ex)
        gen.set_state(st)
        idx = torch.randint(0, (cap - 1) * E, (n,), device=dev, generator=gen)
        step, env = (pos + 1 + idx // E) % cap, idx % E
        assert int((step == pos).sum()) == 0 and set(step.unique().tolist()) == set(range(cap)) - {pos}
        rows = buf.data[step, env]
        assert torch.equal(buf.last_rows, rows) and torch.equal(buf.last_x64[0], rows[:, :nx].double()) and torch.equal(buf.last_x64[1], rows[:, nx: 2 * nx].double())
        row_s, cold_s, row_n, cold_n = buf.last_starts
        nstep = (step + 1) % cap
        cont = (rows[:, -1] == 0) & (nstep != pos)
        rn = torch.where(cont, nstep * E + env, step * E + env)
        ok = buf.iter_ok.reshape(-1)
        assert torch.equal(row_s, step * E + env) and torch.equal(row_n, rn) and int(cont.sum()) > 0 and int((~cont).sum()) > 0
        assert torch.equal(cold_s.bool(), ~ok[row_s]) and torch.equal(cold_n.bool(), ~ok[rn])
    with pytest.raises(RuntimeError):
        buf.full = False
        buf.sample_fused(n, gen, exclude_pos=ex)


def test_td3_pipelined_loop():
    """BatchedTD3(pipeline=True), step() with graphs on one rank: the update runs beside the roll-out step of the same call.  Another
    loop than the sequential one (the sampler leaves out the slot being written, the draws come in another order), so what is asserted
    is what must hold for it on its own: two runs are bit-identical (nothing depends on how the two branches interleave), every
    solve's result is used (finite tables, critics, theta), the write position and the table advance as in the sequential loop, and it
    learns on the same scale (critic loss within a factor of two of the sequential loop's after the same number of steps)."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
    ocp, res = cartpole_ocp(), []
    for pipe in (True, True, False):
        env = BatchedCartPoleSwingUpEnv(512, device="cuda", seed=0, max_episode_steps=9)
        ag = BatchedTD3(ocp, env, batch_size=512, buffer_steps=6, policy_delay=2, lr_actor=1e-4, seed=0, replay_iterates=True, pipeline=pipe)
        ag.collect(2)
        ag.enable_graphs()
        ag.step(14)
        res.append((ag.buffer.data.clone(), torch.cat([p.detach().reshape(-1) for p in ag.critic.parameters()]), ag.theta.clone(), env.state.clone(),
                    ag.buffer.pos_t.clone(), ag.buffer.iter_ok.clone(), ag.last_critic_loss(), (ag.buffer.pos, ag.buffer.full)))
    a, b, c = res
    assert all(torch.equal(x, y) for x, y in zip(a[:6], b[:6])) and a[6] == b[6]
    assert all(bool(torch.isfinite(t).all()) for t in a[:4]) and float((a[2] - torch.as_tensor(ocp.p0, device="cuda")).abs().max()) > 0.0
    assert int(a[4]) == int(c[4]) and a[7] == c[7] and int(a[4]) == a[7][0]
    assert 0.5 < a[6] / c[6] < 2.0
    assert float((a[0] != c[0]).float().mean()) > 0.0      # (it IS another loop)
