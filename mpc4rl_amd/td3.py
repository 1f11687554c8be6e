"""MPC actor for TD3 / deterministic policy gradient: the batched replacement of rlmpc/td3/policies.py:125-222.

The reference's ``Actor.forward`` (186-197) loops ``self._predict(o)`` over the observation batch, each iteration going
torch -> numpy -> ctypes -> numpy -> torch; its actor optimizer is "not being used" (332) and no DPG step exists.
Here ``forward`` is one batched solve, and ``dpg_step`` implements the deterministic policy gradient the stub was heading
for:  grad_theta J = mean_i ( dpi/dtheta_i ' * grad_a Q(s_i, a)|_{a = pi(s_i)} )  with dpi/dtheta from the KKT sensitivities and
grad_a Q from any torch critic.  (No reference behaviour to match for the update rule — SURVEY.md §0-5.)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .batch import MPCBatch
from .distributed import mean_update


class ContinuousCritic(nn.Module):
    """Q(s, a) MLPs as in rlmpc/td3/policies.py:47-122 (n_critics networks on [obs, action])."""

    def __init__(self, obs_dim: int, act_dim: int, net_arch=(64, 64), n_critics: int = 2):
        super().__init__()
        self.q_networks = nn.ModuleList()
        for _ in range(n_critics):
            layers, d = [], obs_dim + act_dim
            for h in net_arch:
                layers += [nn.Linear(d, h), nn.ReLU()]
                d = h
            layers.append(nn.Linear(d, 1))
            self.q_networks.append(nn.Sequential(*layers))

    def forward(self, obs, actions):
        qin = torch.cat([obs, actions], dim=1)
        return tuple(q(qin) for q in self.q_networks)

    def q1_forward(self, obs, actions):
        return self.q_networks[0](torch.cat([obs, actions], dim=1))


class MPCActor:
    """Deterministic policy a = scale(u0*(s; theta)) backed by an MPCBatch (td3/policies.py:125-222)."""

    def __init__(self, ocp, batch: int, device=None, scale: bool = True):
        self.mpc = MPCBatch(ocp, batch, device)
        self.ocp = ocp
        self.low = torch.as_tensor(ocp.lbu, dtype=torch.float64, device=self.mpc.device)
        self.high = torch.as_tensor(ocp.ubu, dtype=torch.float64, device=self.mpc.device)
        self.scale = scale
        self.theta = torch.as_tensor(ocp.p0, dtype=torch.float64, device=self.mpc.device).clone()

    def parameters(self) -> torch.Tensor:                        # Actor.parameters, td3/policies.py:215-222
        return self.theta

    def scale_action(self, u):                                   # mpc.py:290-301
        return 2.0 * ((u - self.low) / (self.high - self.low)) - 1.0 if self.scale else u

    def forward(self, obs: torch.Tensor) -> torch.Tensor:        # Actor.forward, td3/policies.py:186-197 — ONE launch
        return self.scale_action(self.mpc.get_action(obs.to(torch.float64))).to(obs.dtype)

    __call__ = forward

    def dpg_step(self, obs: torch.Tensor, critic: ContinuousCritic, lr: float, group=None) -> torch.Tensor:
        """theta += lr * mean_i( dpi/dtheta_i' grad_a Q_i )  (ascent on Q; use a negative lr for costs)."""
        r = self.mpc.solve(obs.to(torch.float64), sens_pi=True)
        a = self.scale_action(r.u0).to(obs.dtype).detach().requires_grad_(True)
        q = critic.q1_forward(obs, a).sum()
        (dq_da,) = torch.autograd.grad(q, a)
        chain = (2.0 / (self.high - self.low)) if self.scale else torch.ones_like(self.low)
        g = torch.einsum("bu,bup->bp", dq_da.to(torch.float64) * chain, r.dpi_dp)
        ok = (r.status == 0).to(torch.float64)
        step = mean_update(g, lr * ok, group)
        self.theta = self.theta + step
        self.mpc.set_theta(self.theta)
        return step
