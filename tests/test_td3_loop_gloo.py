"""BASELINE config 5 on CPU: the whole batched TD3 loop (roll-out over sharded environments -> replay -> critic TD update -> delayed
deterministic policy gradient -> ONE all-reduce per update) with world_size 2 over gloo.  There is no GPU here, so the MPC actor is
replaced by a closed-form stand-in policy u = -K(theta) x with analytic du/dtheta — the thing under test is the loop's plumbing:
identical critics and identical theta on all ranks after every update, the reference's sampling / target / delay rules
(stable_baselines3 TD3 as driven by scripts/cartpole_mpc_as_td3_agent_closed_loop.py:40-67), per-environment resets."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class StandInMPC:
    """MPCBatch look-alike on the CPU: u0 = clip(-(theta[:4] . x) * 30, +-30); du0/dtheta_j = -30 x_j inside the bounds."""

    def __init__(self, batch, n_p):
        self.B, self.n_p, self.device = batch, n_p, torch.device("cpu")
        self.theta = torch.zeros(n_p, dtype=torch.float64)
        self.calls = []

    def set_theta(self, th):
        self.theta = th.clone()

    def solve(self, x0, sens_pi=False, cold=False, cold_mask=None, **kw):
        self.calls.append((cold, None if cold_mask is None else int(cold_mask.sum())))
        x0 = x0.to(torch.float64)
        raw = -30.0 * (x0 * self.theta[:4]).sum(1, keepdim=True)
        u0 = raw.clamp(-30.0, 30.0)
        dpi = None
        if sens_pi:
            dpi = torch.zeros(x0.shape[0], 1, self.n_p, dtype=torch.float64)
            dpi[:, 0, :4] = -30.0 * x0 * (raw.abs() < 30.0)
        return SimpleNamespace(u0=u0, status=torch.zeros(x0.shape[0], dtype=torch.int32), dpi_dp=dpi)


def _make_actor_factory(ocp):
    from mpc4rl_amd.td3 import MPCActor

    def factory(batch):
        a = MPCActor.__new__(MPCActor)
        a.mpc, a.ocp, a.scale = StandInMPC(batch, ocp.n_p), ocp, True
        a.low, a.high = torch.tensor([-30.0], dtype=torch.float64), torch.tensor([30.0], dtype=torch.float64)
        a.theta = torch.zeros(ocp.n_p, dtype=torch.float64)
        a.theta[:4] = torch.tensor([0.02, 0.05, -0.3, -0.1], dtype=torch.float64)
        a.mpc.set_theta(a.theta)
        return a
    return factory


def _run(rank, world, port, out):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from mpc4rl_amd.envs import BatchedCartPoleSwingUpEnv
    from mpc4rl_amd.problems import cartpole_ocp
    from mpc4rl_amd.td3 import BatchedTD3
    ocp = cartpole_ocp()
    env = BatchedCartPoleSwingUpEnv(16, device="cpu", seed=10 + rank, max_episode_steps=7)      # short episodes: resets happen
    mask = torch.zeros(ocp.n_p, dtype=torch.float64)
    mask[:4] = 1.0
    agent = BatchedTD3(ocp, env, batch_size=32, buffer_steps=24, policy_delay=2, lr_actor=1e-3, seed=3, learn_mask=mask,
                       actor_factory=_make_actor_factory(ocp))
    theta0 = agent.theta.clone()
    st = agent.collect(12)
    assert agent.buffer.size() == 12 * 16 and st["episodes_ended"] >= 16            # every environment was truncated once
    # the roll-out solve after an episode end carried a cold mask for exactly the environments that ended
    assert any(c[1] == 16 for c in agent.actor.mpc.calls) and agent.actor.mpc.calls[0] == (False, None)
    tr = agent.train(4)
    agent.collect(3)
    tr = agent.train(3)
    assert np.isfinite(tr["critic_loss"]) and agent.n_updates == 7
    crit = torch.cat([p.detach().reshape(-1) for p in agent.critic.parameters()]).numpy().copy()
    targ = torch.cat([p.detach().reshape(-1) for p in agent.critic_target.parameters()]).numpy().copy()
    out[rank] = (agent.theta.numpy().copy(), agent.theta_target.numpy().copy(), crit, targ, theta0.numpy().copy(),
                 agent.buffer.obs[:12].numpy().copy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_td3_closed_loop_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), out), nprocs=world, join=True)
    th0, tt0, c0, t0, theta_init, obs0 = out[0]
    th1, tt1, c1, t1, _, obs1 = out[1]
    # every rank holds the same actor parameters, target parameters, critic and critic target — bitwise
    assert np.array_equal(th0, th1) and np.array_equal(tt0, tt1) and np.array_equal(c0, c1) and np.array_equal(t0, t1)
    assert not np.array_equal(obs0, obs1)                                  # ... although they saw different environments
    assert np.abs(th0 - theta_init)[:4].max() > 0.0 and np.all(th0[4:] == theta_init[4:])       # the policy moved, only where allowed
    assert np.abs(tt0 - theta_init).max() > 0.0 and np.abs(tt0 - theta_init).max() < np.abs(th0 - theta_init).max()   # Polyak lag
    assert np.abs(c0 - t0).max() > 0.0


def test_td3_single_rank_matches_formulas():
    """One rank, one update, recomputed by hand: the TD target, the critic loss and the deterministic policy gradient step."""
    out = {}
    _run(0, 1, 0, out)                       # the loop runs without a process group as well
    from mpc4rl_amd.envs import BatchedCartPoleSwingUpEnv
    from mpc4rl_amd.problems import cartpole_ocp
    from mpc4rl_amd.td3 import BatchedTD3
    ocp = cartpole_ocp()
    env = BatchedCartPoleSwingUpEnv(8, device="cpu", seed=1)
    mask = torch.zeros(ocp.n_p, dtype=torch.float64)
    mask[:4] = 1.0
    ag = BatchedTD3(ocp, env, batch_size=16, buffer_steps=8, policy_delay=1, lr_actor=1e-3, target_noise=0.0, seed=5, learn_mask=mask,
                    actor_factory=_make_actor_factory(ocp))
    ag.collect(8)
    gen_state = ag.gen.get_state()
    crit0 = [p.detach().clone() for p in ag.critic.parameters()]
    theta0 = ag.theta.clone()
    ag.train(1)
    # replay the same sample by hand
    ag.gen.set_state(gen_state)
    obs, nxt, act, rew, done = ag.buffer.sample(16, ag.gen)
    import copy
    ref = copy.deepcopy(ag.critic)
    for p, p0 in zip(ref.parameters(), crit0):
        p.data.copy_(p0)
    tgt = copy.deepcopy(ref)                 # before the first update the target equals the critic
    a_next = ag.target_mpc.scale_action(ag.target_mpc.mpc.solve(nxt).u0).float().clamp(-1, 1)
    with torch.no_grad():
        y = rew + 0.99 * (1 - done) * torch.min(*tgt(nxt, a_next)).squeeze(1)
    r = StandInMPC(16, ocp.n_p)
    r.set_theta(theta0)
    rp = r.solve(obs, sens_pi=True)
    a_pi = (2.0 * (rp.u0 + 30.0) / 60.0 - 1.0).float().requires_grad_(True)
    (dq,) = torch.autograd.grad(ref.q1_forward(obs, a_pi).sum(), a_pi)
    g = torch.einsum("bu,bup->bp", dq.double() * (2.0 / 60.0), rp.dpi_dp).mean(0)
    assert torch.allclose(ag.theta - theta0, 1e-3 * mask * g, rtol=1e-9, atol=1e-15)
    loss = sum(((q.squeeze(1) - y) ** 2).mean() for q in ref(obs, act))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    loss.backward()
    opt.step()
    for p, q in zip(ag.critic.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
