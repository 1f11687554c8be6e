"""CPU tests of the rows either side of the hot path: batched environments and the YAML loader."""
import math
import os

import numpy as np
import torch


def test_batched_cartpole_env_matches_reference_step_formulas():
    """Per-environment restatement of ContinuousCartPoleSwingUpEnv.step (continuous_cartpole/environment.py:105-134)."""
    from mpc4rl_amd import BatchedCartPoleSwingUpEnv
    env = BatchedCartPoleSwingUpEnv(16, seed=3)
    s0 = env.reset().numpy()
    assert np.all(s0[:, [0, 1, 3]] == 0.0) and np.all((s0[:, 2] >= 0.9 * math.pi) & (s0[:, 2] <= 1.1 * math.pi))
    rng = np.random.default_rng(0)
    state = s0.copy()
    for _ in range(25):
        act = rng.uniform(-1, 1, (16, 1))
        obs, rew, term, trunc = env.step(torch.as_tensor(act))
        nxt = np.zeros_like(state)
        for i in range(16):
            x, x_dot, theta, theta_dot = state[i]
            force = act[i, 0] * 30.0
            costheta, sintheta = math.cos(theta), math.sin(theta)
            temp = (force + 0.05 * theta_dot ** 2 * sintheta) / 1.1
            thetaacc = (9.8 * sintheta - costheta * temp) / (0.5 * (4.0 / 3.0 - 0.1 * costheta ** 2 / 1.1))
            xacc = temp - 0.05 * thetaacc * costheta / 1.1
            nxt[i] = [x + 0.02 * x_dot, x_dot + 0.02 * xacc, theta + 0.02 * theta_dot, theta_dot + 0.02 * thetaacc]
        assert np.allclose(obs.numpy(), nxt, rtol=1e-13, atol=1e-13)
        assert np.allclose(rew.numpy(), nxt[:, 0] ** 2 + nxt[:, 2] ** 2)
        assert not term.any()
        state = nxt
    env.state[0] = torch.tensor([0.05, 0.01, 0.01, 0.02], dtype=torch.float64)
    assert bool(env.is_terminal(env.state)[0])
    mask = torch.zeros(16, dtype=torch.bool)
    mask[3] = True
    before = env.state.clone()
    env.reset(mask)
    assert torch.equal(env.state[:3], before[:3]) and env.state[3, 0] == 0.0
    # reset_where: the same masked reset without reading the number of ended environments back
    env.steps[:] = 7
    mask[5] = True
    before = env.state.clone()
    o = env.reset_where(mask)
    keep = ~mask
    assert torch.equal(o, env.state) and torch.equal(o[keep], before[keep]) and bool((env.steps[keep] == 7).all()) and bool((env.steps[mask] == 0).all())
    assert bool((o[mask][:, [0, 1, 3]] == 0).all()) and bool((o[mask][:, 2] >= 0.9 * np.pi).all()) and bool((o[mask][:, 2] < 1.1 * np.pi).all())


def test_batched_linear_env_matches_reference():
    """rlmpc/gym/linear_system/environment.py:28-66."""
    from mpc4rl_amd import BatchedLinearSystemEnv
    env = BatchedLinearSystemEnv(8, lb_noise=-0.1, ub_noise=0.1, seed=1)
    s = env.reset()
    assert torch.equal(s, torch.tensor([[0.5, 0.5]] * 8, dtype=torch.float64))
    a = torch.linspace(-1, 1, 8, dtype=torch.float64).reshape(8, 1)
    obs, cost, done, _ = env.step(a)
    A, B = np.array([[0.9, 0.35], [0.0, 1.1]]), np.array([[0.0813], [0.2]])
    det = s.numpy() @ A.T + a.numpy() @ B.T
    noise = obs.numpy() - det
    assert np.all(noise[:, 1] == 0.0) and np.all((noise[:, 0] >= -0.1) & (noise[:, 0] <= 0.1))
    o = obs.numpy()
    viol_lo = ((np.array([0.0, -1.0]) - o).clip(min=0) > 0).any(1) * 1e2
    viol_hi = ((o - np.array([1.0, 1.0])).clip(min=0) > 0).any(1) * 1e2
    assert np.allclose(cost.numpy(), 0.5 * (o * o).sum(1) + 0.5 * (a.numpy() ** 2).sum(1) + viol_lo + viol_hi)
    assert not done.any()


def test_yaml_loader_matches_reference_config_layout():
    from mpc4rl_amd import cartpole_ocp, cartpole_ocp_from_config, read_config
    cfg = read_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cartpole_config.yaml"))
    o = cartpole_ocp_from_config(cfg)
    assert o.N == 30 and abs(o.dT - 0.1) < 1e-15 and abs(o.h - 0.025) < 1e-15      # config/cartpole.yaml:7,11,23
    ref = cartpole_ocp(N=30, tf=3.0)
    assert np.allclose(o.p0, ref.p0) and np.allclose(o.consts, ref.consts)
    for a, b in zip(o.stage_bounds(), ref.stage_bounds()):
        assert np.array_equal(a, b)
    assert np.allclose(o.x0, [0.0, 0.0, 3.14, 0.0]) and o.max_iter == 500
    # a stage-0 cost of its own (W_0 != W, yref_0 != yref: the yaml has separate keys, cartpole/acados.py:111-186) lands in the W_0 /
    # yref_0 fields of p, which the kernels read (csrc/models_dev.hpp CartpoleDev::P_W0, P_YREF0)
    import copy
    cfg2 = copy.deepcopy(cfg)
    cfg2["mpc"]["cost"]["W_0"][0][0] = 3.0
    cfg2["mpc"]["cost"]["yref_0"][2] = 0.25
    o2 = cartpole_ocp_from_config(cfg2)
    f = o2.cost_fields
    W0 = o2.p0[f["W_0"][0]: f["W_0"][0] + 25].reshape(5, 5, order="F")
    assert W0[0, 0] == 3.0 and np.allclose(W0[1:, 1:], np.asarray(cfg["mpc"]["cost"]["W"])[1:, 1:])
    assert o2.p0[f["yref_0"][0] + 2] == 0.25 and np.allclose(o2.p0[f["W"][0]: f["W"][0] + 25], o.p0[f["W"][0]: f["W"][0] + 25])
