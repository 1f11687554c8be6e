"""Batched torch environments: the step either side of the MPC hot path, producing x0[B, nx] on the device.

Restates (no gymnasium dependency, tensors in / tensors out, B environments per call):
  * ContinuousCartPoleSwingUpEnv   rlmpc/gym/continuous_cartpole/environment.py:19-200
      Euler, tau = 0.02, force = action * force_mag (30), gravity 9.8, masscart 1.0, masspole 0.1, length 0.5;
      reset: theta ~ U(0.9 pi, 1.1 pi), other states 0 (178-180); reward = x^2 + theta^2 (193-194);
      terminated when |x| < 0.1, |x_dot| < 0.1, |theta| < 2 deg, |theta_dot| < 0.1 all hold (84-103)
  * LinearSystemEnv                rlmpc/gym/linear_system/environment.py:6-66
      s+ = A s + B a + [U(lb_noise, ub_noise), 0]; reset to [0.5, 0.5]; cost = 1/2 s's + 1/2 a'a + 100 per violated side
The STATE is always float64 (the reference's numpy state); observations are returned in ``dtype`` — float64 by default (they feed the
fp64 solve directly), ``dtype=torch.float32`` for what the reference's gymnasium envs return (``np.array(self.state, dtype=np.float32)``,
continuous_cartpole/environment.py:166,186).  On a GPU every call is ONE launch of the library's environment kernels
(csrc/env_kernel.hpp, which store the observation in either type); the torch formulation below them is the CPU form (gloo tests) and
updates the same tensors in place.
The reference's own numpy VectorEnv (environment.py:302-458) computes the reward from env 0's action and resets to
exactly [0, 0, pi, 0]; the single-env semantics above are the ones reproduced here, per environment.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


class BatchedCartPoleSwingUpEnv:
    def __init__(self, num_envs: int, device="cpu", force_mag: float = 30.0, max_episode_steps: int = 500, seed: int = 0,
                 dtype=torch.float64):
        self.num_envs, self.device, self.dtype = num_envs, torch.device(device), dtype
        self.gravity, self.masscart, self.masspole, self.length = 9.8, 1.0, 0.1, 0.5
        self.total_mass = self.masspole + self.masscart
        self.polemass_length = self.masspole * self.length
        self.force_mag, self.tau = force_mag, 0.02
        self.x_threshold, self.theta_threshold = 0.1, math.radians(2.0)
        self.max_episode_steps = max_episode_steps
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("observations are float64 or float32")
        self.state = torch.zeros(num_envs, 4, dtype=torch.float64, device=self.device)     # never rebound: graphs hold its address
        self.steps = torch.zeros(num_envs, dtype=torch.int64, device=self.device)
        self._par_c = None

    def _sample(self, n: int) -> torch.Tensor:
        s = torch.zeros(n, 4, dtype=torch.float64, device=self.device)
        s[:, 2] = (0.9 + 0.2 * torch.rand(n, generator=self.gen, dtype=torch.float64, device=self.device)) * math.pi
        return s

    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is None:
            self.state.copy_(self._sample(self.num_envs))
            self.steps.zero_()
        else:
            n = int(mask.sum().item())
            if n:
                self.state[mask] = self._sample(n)
                self.steps[mask] = 0
        return self.state.to(self.dtype, copy=True)

    # ---- device path: one launch per call through the C ABI (mpcrl_env_cartpole_step / _reset, csrc/env_kernel.hpp) ----
    def _native(self) -> bool:
        return self.device.type == "cuda"

    def _f32(self) -> int:
        return 1 if self.dtype == torch.float32 else 0

    def _par(self):
        # rebuilt on every call (nine host scalars): an edit of env.force_mag, max_episode_steps, ... after construction reaches the
        # kernel exactly as it reaches the torch formulation below
        import ctypes
        self._par_c = (ctypes.c_double * 9)(self.gravity, self.masscart, self.masspole, self.length, self.force_mag, self.tau,
                                            self.x_threshold, self.theta_threshold, float(self.max_episode_steps))
        return self._par_c

    def reset_where(self, mask: torch.Tensor) -> torch.Tensor:
        """reset(mask) without a host synchronisation (no count of the ended environments is read back): B uniform numbers are
        drawn and the masked environments take theirs.  Returns the observation of every environment after the reset."""
        u01 = torch.rand(self.num_envs, generator=self.gen, dtype=torch.float64, device=self.device)
        mask = mask.to(self.device)
        if self._native():
            from . import _lib
            from .batch import _ptr
            m8 = mask.to(torch.uint8).contiguous()
            obs = torch.empty(self.num_envs, 4, dtype=self.dtype, device=self.device)
            rc = _lib.load().mpcrl_env_cartpole_reset(self.num_envs, _ptr(self.state), _ptr(self.steps), _ptr(m8), _ptr(u01), _ptr(obs),
                                                      self._f32(), torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"mpcrl_env_cartpole_reset failed with code {rc}")
            return obs
        fresh = torch.zeros_like(self.state)
        fresh[:, 2] = (0.9 + 0.2 * u01) * math.pi
        self.state.copy_(torch.where(mask[:, None], fresh, self.state))
        self.steps.copy_(torch.where(mask, torch.zeros_like(self.steps), self.steps))
        return self.state.to(self.dtype, copy=True)

    def is_terminal(self, s: torch.Tensor) -> torch.Tensor:
        return (s[:, 0].abs() < self.x_threshold) & (s[:, 1].abs() < 0.1) & (s[:, 2].abs() < self.theta_threshold) & (s[:, 3].abs() < 0.1)

    def step(self, action: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """action: [B, 1] in [-1, 1] (what CartpoleMPC.get_action returns).  Returns obs, reward, terminated, truncated."""
        a = action.to(torch.float64).reshape(self.num_envs)
        if self._native():
            from . import _lib
            from .batch import _ptr
            a = a.to(self.device).contiguous()
            obs = torch.empty(self.num_envs, 4, dtype=self.dtype, device=self.device)
            reward = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
            flags = torch.empty((2, self.num_envs), dtype=torch.uint8, device=self.device)
            rc = _lib.load().mpcrl_env_cartpole_step(self._par(), self.num_envs, _ptr(self.state), _ptr(self.steps), _ptr(a), _ptr(obs),
                                                     self._f32(), _ptr(reward), _ptr(flags[0]), _ptr(flags[1]),
                                                     torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"mpcrl_env_cartpole_step failed with code {rc}")
            fb = flags.view(torch.bool)
            return obs, reward, fb[0], fb[1]
        x, x_dot, th, th_dot = self.state.unbind(1)
        force = a * self.force_mag
        c, s = torch.cos(th), torch.sin(th)
        temp = (force + self.polemass_length * th_dot ** 2 * s) / self.total_mass
        thacc = (self.gravity * s - c * temp) / (self.length * (4.0 / 3.0 - self.masspole * c ** 2 / self.total_mass))
        xacc = temp - self.polemass_length * thacc * c / self.total_mass
        self.state.copy_(torch.stack([x + self.tau * x_dot, x_dot + self.tau * xacc, th + self.tau * th_dot, th_dot + self.tau * thacc], 1))
        self.steps += 1
        terminated = self.is_terminal(self.state)
        reward = self.state[:, 0] ** 2 + self.state[:, 2] ** 2
        truncated = self.steps >= self.max_episode_steps
        return self.state.to(self.dtype, copy=True), reward, terminated, truncated


class BatchedLinearSystemEnv:
    def __init__(self, num_envs: int, device="cpu", lb_noise: float = -0.1, ub_noise: float = 0.0, A=None, B=None,
                 min_observation=(-0.0, -1.0), max_observation=(1.0, 1.0), seed: int = 0, dtype=torch.float64):
        self.num_envs, self.device, self.dtype = num_envs, torch.device(device), dtype
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("observations are float64 or float32")
        kw = dict(dtype=torch.float64, device=self.device)                              # state and parameters: fp64 (numpy state)
        self.A = torch.tensor([[0.9, 0.35], [0.0, 1.1]] if A is None else A, **kw)     # environment.py:15
        self.B = torch.tensor([[0.0813], [0.2]] if B is None else B, **kw)
        self.lb_noise, self.ub_noise = lb_noise, ub_noise
        self.low, self.high = torch.tensor(min_observation, **kw), torch.tensor(max_observation, **kw)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.state = torch.zeros(num_envs, 2, **kw)
        self._par_c, self._par_key = None, None

    def reset(self) -> torch.Tensor:
        self.state.copy_(torch.tensor([0.5, 0.5], dtype=torch.float64, device=self.device).repeat(self.num_envs, 1))   # environment.py:46
        return self.state.to(self.dtype, copy=True)

    def cost(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        lower = ((self.low - state).clamp(min=0) > 0).any(1).to(torch.float64) * 1e2
        upper = ((state - self.high).clamp(min=0) > 0).any(1).to(torch.float64) * 1e2
        return 0.5 * (state * state).sum(1) + 0.5 * (action * action).sum(1) + lower + upper

    def step(self, action: torch.Tensor):
        a = action.to(torch.float64).reshape(self.num_envs, 1)
        if self.device.type == "cuda":   # one launch through the C ABI (csrc/env_kernel.hpp)
            import ctypes
            from . import _lib
            from .batch import _ptr
            # the parameter block is cached against the identity and in-place version of the tensors it was read from (and the noise
            # bounds), so that a later edit of env.A / env.B / env.low / env.high / the noise bounds is honoured like on the torch path
            # without a device-to-host copy per step
            key = (id(self.A), self.A._version, id(self.B), self.B._version, id(self.low), self.low._version, id(self.high),
                   self.high._version, float(self.lb_noise), float(self.ub_noise))
            if self._par_c is None or self._par_key != key:
                vals = self.A.reshape(-1).tolist() + self.B.reshape(-1).tolist() + [self.lb_noise, self.ub_noise] + self.low.tolist() + self.high.tolist()
                self._par_c, self._par_key = (ctypes.c_double * 12)(*vals), key
            u01 = torch.rand(self.num_envs, generator=self.gen, dtype=torch.float64, device=self.device)
            a = a.to(self.device).contiguous()
            obs = torch.empty(self.num_envs, 2, dtype=self.dtype, device=self.device)
            cost = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
            rc = _lib.load().mpcrl_env_linear_step(self._par_c, self.num_envs, _ptr(self.state), _ptr(a), _ptr(u01), _ptr(obs),
                                                   1 if self.dtype == torch.float32 else 0, _ptr(cost),
                                                   torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"mpcrl_env_linear_step failed with code {rc}")
            done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
            return obs, cost, done, done.clone()
        noise = torch.zeros(self.num_envs, 2, dtype=torch.float64, device=self.device)
        noise[:, 0] = self.lb_noise + (self.ub_noise - self.lb_noise) * torch.rand(self.num_envs, generator=self.gen, dtype=torch.float64,
                                                                                      device=self.device)
        self.state.copy_(self.state @ self.A.T + a @ self.B.T + noise)
        reward = self.cost(self.state, a)
        done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        return self.state.to(self.dtype, copy=True), reward, done, done.clone()
