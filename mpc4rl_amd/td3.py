"""MPC actor for TD3 / deterministic policy gradient: the batched replacement of rlmpc/td3/policies.py:125-222.

The reference's ``Actor.forward`` (186-197) loops ``self._predict(o)`` over the observation batch, each iteration going
torch -> numpy -> ctypes -> numpy -> torch; its actor optimizer is "not being used" (332) and no DPG step exists.
Here ``forward`` is one batched solve, and ``dpg_step`` implements the deterministic policy gradient the stub was heading
for:  grad_theta J = mean_i ( dpi/dtheta_i ' * grad_a Q(s_i, a)|_{a = pi(s_i)} )  with dpi/dtheta from the KKT sensitivities and
grad_a Q from any torch critic.  (No reference behaviour to match for the update rule — SURVEY.md §0-5.)
"""
from __future__ import annotations

import ctypes as C
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


def flatten_parameters(module: nn.Module):
    """(flat parameters, flat gradients): every parameter of ``module`` becomes a view into ONE contiguous float32 tensor, in
    ``module.parameters()`` order, and gets a persistent ``.grad`` that is a view into a second one — the layout mpcrl_critic_td_grad
    reads and writes.  In-place updates (fused Adam, load_state_dict, Polyak averaging) keep the views; nothing may rebind ``p.data``."""
    params = list(module.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    grad = torch.zeros_like(flat)
    off = 0
    for p in params:
        n = p.numel()
        p.data = flat[off: off + n].view(p.shape)
        p.grad = grad[off: off + n].view(p.shape)
        off += n
    return flat, grad


def _as_u8(mask: torch.Tensor) -> torch.Tensor:
    """a [B] mask as contiguous uint8 without a launch where it is bool already (one byte per entry, 0 / 1)"""
    m = mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else (m if m.dtype == torch.uint8 else m.to(torch.uint8))


def critic_td_grad(rows, nx: int, nu: int, a_next, ok_u, params, params_target, n_critics: int, gamma: float, out_scale: float, grad_out,
                   workspace=None):
    """mpcrl_critic_td_grad (include/mpcrl.h): TD target, twin-critic loss and its gradient with respect to the flat critic parameters —
    two launches.  rows [B, >= 2 nx + nu + 2] float32 (obs | next obs | action | reward | done), a_next [B, nu] float32, ok_u [B] bool /
    uint8 or None; grad_out: float64 [n_params] (written).  Returns (loss [1] float32, ok [B] bool, workspace)."""
    from . import _lib
    lib = _lib.load()
    B = rows.shape[0]
    dev = rows.device
    if not (rows.dtype == torch.float32 and rows.stride(1) == 1 and a_next.dtype == torch.float32 and a_next.is_contiguous()):
        raise ValueError("rows / a_next: float32, unit inner stride")
    need = lib.mpcrl_critic_workspace_bytes(B, nx, nu, n_critics)
    if need < 0:
        raise ValueError("mpcrl_critic_td_grad: nx + nu <= 64, n_critics 1 or 2")
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=dev)
    okb = None if ok_u is None else _as_u8(ok_u)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ok = torch.empty(B, dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        rc = lib.mpcrl_critic_td_grad(ptr(rows), rows.stride(0), B, nx, nu, ptr(a_next), ptr(okb), ptr(params), ptr(params_target), n_critics,
                                      float(gamma), float(out_scale), ptr(workspace), ptr(grad_out), ptr(loss), ptr(ok),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"mpcrl_critic_td_grad failed with code {rc}")
    return loss, ok.view(torch.bool), workspace


def critic_dq_da(obs, nx: int, act, ok_u, params):
    """mpcrl_critic_dq_da: dQ_1/da at (obs[:, :nx], act), one launch; 0 where ok_u is 0 or an input is not finite.  Returns (dq_da [B, nu]
    float32, ok [B] bool)."""
    from . import _lib
    B, nu = act.shape
    dev = act.device
    if not (obs.dtype == torch.float32 and obs.stride(1) == 1 and act.dtype == torch.float32 and act.is_contiguous()):
        raise ValueError("obs / act: float32, unit inner stride")
    okb = None if ok_u is None else _as_u8(ok_u)
    out = torch.empty((B, nu), dtype=torch.float32, device=dev)
    ok = torch.empty(B, dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        rc = _lib.load().mpcrl_critic_dq_da(ptr(obs), obs.stride(0), B, nx, nu, ptr(act), ptr(okb), ptr(params), ptr(out), ptr(ok),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"mpcrl_critic_dq_da failed with code {rc}")
    return out, ok.view(torch.bool)


def dpg_grad(dq_da, ok, dpi_dp, low, high, scale: bool, out, workspace=None):
    """mpcrl_dpg_grad: out[:n_p] = sum_b ok_b sum_u dq_da[b, u] chain_u dpi_dp[b, u, :], out[n_p] = sum_b ok_b — one launch, fixed summation
    order.  out: float64 [n_p + 1] (written; may be a slice of a larger message).  Returns the workspace to pass again."""
    from . import _lib
    lib = _lib.load()
    B, nu, n_p = dpi_dp.shape
    dev = dpi_dp.device
    if not (dq_da.dtype == torch.float32 and dq_da.is_contiguous() and dpi_dp.dtype == torch.float64 and dpi_dp.is_contiguous() and out.is_contiguous()):
        raise ValueError("dq_da: contiguous float32 [B, nu]; dpi_dp, out: contiguous float64")
    need = lib.mpcrl_dpg_workspace_bytes(B, n_p)
    if workspace is None or workspace.numel() * 8 < need:
        workspace = torch.zeros((need + 7) // 8, dtype=torch.float64, device=dev)       # (zero once: the kernel leaves its counter zero)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        rc = lib.mpcrl_dpg_grad(ptr(dq_da), ptr(None if ok is None else _as_u8(ok)), ptr(dpi_dp), B, nu, n_p, ptr(low), ptr(high), int(scale),
                                ptr(workspace), ptr(out), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"mpcrl_dpg_grad failed with code {rc}")
    return workspace


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

    def action(self, r, noise: Optional[torch.Tensor] = None, sigma: float = 0.0, noise_clip: float = 0.0, accept_status2: bool = False):
        """(a [B, nu] float32, ok [B] bool): the policy's action of a solve result — scale_action where the solve succeeded (status 0, with
        accept_status2 also 2: the closed loop applies what max_iter left) and u0 is finite, 0 elsewhere — plus
        clip(sigma noise, +-noise_clip), the sum clipped to [-1, 1] when ``noise`` (standard-normal draws, float32) is given.  ONE launch
        (mpcrl_policy_action) for what is ~15 elementwise launches in torch; the same arithmetic in the same order and types."""
        from . import _lib
        B, nu = r.u0.shape
        a = torch.empty((B, nu), dtype=torch.float32, device=r.u0.device)
        ok = torch.empty((B,), dtype=torch.uint8, device=r.u0.device)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        if noise is not None and not (noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == (B, nu)):
            raise ValueError("noise: contiguous float32 [B, nu]")
        with torch.cuda.device(r.u0.device):
            rc = _lib.load().mpcrl_policy_action(ptr(r.u0), ptr(r.status), ptr(noise), ptr(self.low), ptr(self.high), B, nu, int(self.scale), float(sigma),
                                                 float(noise_clip), int(accept_status2), ptr(a), ptr(ok), C.c_void_p(torch.cuda.current_stream(r.u0.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"mpcrl_policy_action failed with code {rc}")
        return a, ok.view(torch.bool)        # (0 / 1 bytes: a view, not a launch)

    def forward(self, obs: torch.Tensor) -> torch.Tensor:        # Actor.forward, td3/policies.py:186-197 — ONE launch
        return self.scale_action(self.mpc.get_action(obs.to(torch.float64))).to(obs.dtype)

    __call__ = forward

    def dpg_step(self, obs: torch.Tensor, critic: ContinuousCritic, lr: float, group=None) -> torch.Tensor:
        """theta += lr * mean_i( dpi/dtheta_i' grad_a Q_i )  (ascent on Q; use a negative lr for costs)."""
        r = self.mpc.solve(obs.to(torch.float64), sens_pi=True)
        okb = (r.status == 0) & torch.isfinite(r.u0).all(dim=1)
        u = torch.where(okb[:, None], torch.nan_to_num(r.u0), torch.zeros_like(r.u0))   # select, never multiply by a mask: NaN * 0 = NaN
        a = self.scale_action(u).to(obs.dtype).detach().requires_grad_(True)
        q = critic.q1_forward(obs, a).sum()
        (dq_da,) = torch.autograd.grad(q, a)
        chain = (2.0 / (self.high - self.low)) if self.scale else torch.ones_like(self.low)
        g = torch.einsum("bu,bup->bp", dq_da.to(torch.float64) * chain, torch.nan_to_num(r.dpi_dp))
        g = torch.where(okb[:, None], g, torch.zeros_like(g))
        ok = okb.to(torch.float64)
        step = mean_update(g, lr * ok, group)
        self.theta = self.theta + step
        self.mpc.set_theta(self.theta)
        return step


class MPCTD3Policy:
    """The shape of rlmpc/td3/policies.py:225-361 (``MPCTD3Policy(BasePolicy)``) without stable-baselines3 (not installable here): the
    attributes an SB3 ``TD3`` algorithm reaches for — ``actor``, ``actor_target``, ``critic``, ``critic_target`` with their
    ``optimizer``s — and the methods it calls (``make_actor``, ``make_critic``, ``forward``, ``_predict``, ``set_training_mode``), with
    the reference's argument names.  ``observation_space`` / ``action_space`` only need ``.shape`` (gymnasium spaces or stand-ins);
    ``lr_schedule`` is a callable of the remaining progress (``lr_schedule(1)`` at construction, policies.py:322,342); ``mpc`` is an
    ``OcpDescription`` (the reference passes its MPC object: here the actor owns a batched handle built from the same description).
    ``forward(obs)`` is ONE batched solve where the reference's Actor.forward loops ``_predict`` over the batch (policies.py:186-197).
    The actor's "parameters" are the MPC's theta (policies.py:215-222); its optimiser step is ``MPCActor.dpg_step``."""

    def __init__(self, observation_space, action_space, lr_schedule, mpc, net_arch=None, n_critics: int = 1, batch: int = 1, device=None,
                 optimizer_class=torch.optim.Adam, optimizer_kwargs=None, share_features_extractor: bool = False):
        self.observation_space, self.action_space = observation_space, action_space
        self.net_arch = [400, 300] if net_arch is None else list(net_arch)          # policies.py:288-292
        self.optimizer_class, self.optimizer_kwargs = optimizer_class, dict(optimizer_kwargs or {})
        self.actor_kwargs = {"mpc": mpc, "batch": batch, "device": device}
        self.critic_kwargs = {"n_critics": n_critics, "net_arch": self.net_arch}
        self.share_features_extractor = share_features_extractor                  # (observations are states: nothing to extract)
        self.training = True
        self._build(lr_schedule)

    def _build(self, lr_schedule) -> None:                                        # policies.py:318-353
        self.actor = self.make_actor()
        self.actor_target = self.make_actor()
        self.actor.lr = float(lr_schedule(1))                                     # (the reference builds an optimiser it never steps, policies.py:332)
        self.critic = self.make_critic()
        self.critic_target = self.make_critic()
        self.critic_target.load_state_dict(self.critic.state_dict())
        self.critic.optimizer = self.optimizer_class(self.critic.parameters(), lr=float(lr_schedule(1)), **self.optimizer_kwargs)
        self.critic_target.train(False)

    def make_actor(self, features_extractor=None) -> "MPCActor":                  # policies.py:355-357
        kw = self.actor_kwargs
        return MPCActor(kw["mpc"], kw["batch"], kw["device"])

    def make_critic(self, features_extractor=None) -> ContinuousCritic:           # policies.py:359-361
        obs_dim, act_dim = int(self.observation_space.shape[0]), int(self.action_space.shape[0])
        critic = ContinuousCritic(obs_dim, act_dim, self.critic_kwargs["net_arch"], self.critic_kwargs["n_critics"])
        return critic.to(self.actor.mpc.device if hasattr(self, "actor") else "cpu")

    def forward(self, observation: torch.Tensor, deterministic: bool = False) -> torch.Tensor:
        return self._predict(observation, deterministic=deterministic)

    def _predict(self, observation: torch.Tensor, deterministic: bool = False) -> torch.Tensor:
        return self.actor(observation)                                            # a deterministic policy either way

    __call__ = forward

    def set_training_mode(self, mode: bool) -> None:
        self.critic.train(mode)
        self.training = mode


class DeviceReplayBuffer:
    """Ring buffer of transitions on the device, filled E environments at a time (stable_baselines3's ReplayBuffer as
    scripts/cartpole_mpc_as_td3_agent_closed_loop.py:47-58 uses it: obs, next_obs, action, reward, done)."""

    def __init__(self, capacity_steps: int, num_envs: int, obs_dim: int, act_dim: int, device, dtype=torch.float32, iterate_dims=None):
        """iterate_dims = (len x, len u, len pi, len bnd) per environment: the buffer also keeps, per transition, the solver iterate the
        roll-out policy ended with at ``obs`` (float64; BatchedTD3(replay_iterates=True)) and whether that solve converged."""
        kw = dict(dtype=dtype, device=device)
        self.cap, self.E = capacity_steps, num_envs
        self.iters = None
        if iterate_dims is not None:
            # flat tables [capacity_steps * num_envs, len]: row = step * num_envs + env (what mpcrl_get/set_iterate_rows index)
            self.iters = [torch.zeros(capacity_steps * num_envs, n, dtype=torch.float64, device=device) for n in iterate_dims]
            self.iter_ok = torch.zeros(capacity_steps, num_envs, dtype=torch.bool, device=device)
            self._env_ids = torch.arange(num_envs, dtype=torch.int64, device=device)
        # ONE packed table [capacity_steps, num_envs, obs | next_obs | act | rew | done]: a transition is written by one index_copy and a
        # batch is sampled by one gather (five of each before round 6: ~4.5 us per launch inside a replayed graph); obs / next_obs / act /
        # rew / done are views of it
        self._nx, self._nu = obs_dim, act_dim
        self.data = torch.zeros(capacity_steps, num_envs, 2 * obs_dim + act_dim + 2, **kw)
        self.pos, self.full = 0, False
        self.pos_t = torch.zeros(1, dtype=torch.int64, device=device)      # the write position as the device sees it (graph replays)

    obs = property(lambda self: self.data[..., : self._nx])
    next_obs = property(lambda self: self.data[..., self._nx: 2 * self._nx])
    act = property(lambda self: self.data[..., 2 * self._nx: 2 * self._nx + self._nu])
    rew = property(lambda self: self.data[..., -2])
    done = property(lambda self: self.data[..., -1])

    def _row(self, obs, next_obs, act, rew, done) -> torch.Tensor:
        dt = self.data.dtype
        return torch.cat([obs.to(dt), next_obs.to(dt), act.to(dt), rew.to(dt)[:, None], done.to(dt)[:, None]], dim=1)

    def add(self, obs, next_obs, act, rew, done, iterate=None, iterate_ok=None) -> None:
        i = self.pos
        self.data[i] = self._row(obs, next_obs, act, rew, done)
        if self.iters is not None:
            iterate.get_iterate_rows(*self.iters, index=i * self.E + self._env_ids)      # rows of slot i := the handle's stored iterates
            self.iter_ok[i] = iterate_ok
        self._advance()
        if self.iters is not None:
            self.pos_t.fill_(self.pos)       # (the device-side copy of the write position: iterates_of_last_sample reads it)

    def _advance(self) -> None:
        self.pos = (self.pos + 1) % self.cap
        self.full = self.full or self.pos == 0

    def add_at_device_pos(self, obs, next_obs, act, rew, done, iterate=None, iterate_ok=None) -> None:
        """add() with the position read from ``pos_t`` on the device: the identical launches whatever the position, so that a captured
        roll-out step can be replayed.  The host mirror (pos, full) is advanced by the caller once per replay (_advance)."""
        if self.iters is not None:
            iterate.get_iterate_rows(*self.iters, index=self.pos_t * self.E + self._env_ids)
            self.iter_ok.index_copy_(0, self.pos_t, iterate_ok[None])
        self.data.index_copy_(0, self.pos_t, self._row(obs, next_obs, act, rew, done)[None])
        self.pos_t.add_(1).remainder_(self.cap)

    def size(self) -> int:
        return (self.cap if self.full else self.pos) * self.E

    def sample(self, n: int, gen: Optional[torch.Generator] = None):
        steps = self.cap if self.full else self.pos
        if steps == 0:
            raise RuntimeError("DeviceReplayBuffer.sample: the buffer is empty (call collect() / add() first)")
        idx = torch.randint(0, steps * self.E, (n,), device=self.data.device, generator=gen)
        rows = self.data.reshape(self.cap * self.E, -1)[idx]
        self.last_idx, self.last_steps, self.last_rows = idx, steps, rows
        nx, nu = self._nx, self._nu
        return rows[:, :nx], rows[:, nx: 2 * nx], rows[:, 2 * nx: 2 * nx + nu], rows[:, -2], rows[:, -1]

    def sample_fused(self, n: int, gen: Optional[torch.Generator] = None, exclude_pos: Optional[torch.Tensor] = None):
        """sample() + iterates_of_last_sample() + the float64 copies of obs / next_obs the replay solves read, through
        mpcrl_replay_sample: the same draw (torch.randint on ``gen``), then ONE launch.  Returns (obs, next_obs, act, rew, done) as
        sample() does — views of ``last_rows`` — and sets ``last_x64`` = (obs, next_obs) in float64 and, with iterate tables,
        ``last_starts`` = (rows for obs, cold mask, rows for next_obs, cold mask), masks as int32 (1 = start cold)."""
        from . import _lib
        steps = self.cap if self.full else self.pos
        if steps == 0:
            raise RuntimeError("DeviceReplayBuffer.sample: the buffer is empty (call collect() / add() first)")
        dev = self.data.device
        # exclude_pos [1] int64 (a full table): that slot is being written by a roll-out that runs beside this update — the draw is
        # over the rows of the other slots and the slot stands in for the write position in the choice of the iterate for next_obs
        if exclude_pos is not None and not (self.full and self.cap >= 2):
            raise RuntimeError("DeviceReplayBuffer.sample_fused(exclude_pos): the table must be full")
        idx = torch.randint(0, (steps - (exclude_pos is not None)) * self.E, (n,), device=dev, generator=gen)
        L, nx, nu = self.data.shape[-1], self._nx, self._nu
        rows = torch.empty((n, L), dtype=torch.float32, device=dev)
        x64 = torch.empty((2, n, nx), dtype=torch.float64, device=dev)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        st = None
        if self.iters is not None:
            r64 = torch.empty((2, n), dtype=torch.int64, device=dev)
            c32 = torch.empty((2, n), dtype=torch.int32, device=dev)
            st = (r64[0], c32[0], r64[1], c32[1])
        with torch.cuda.device(dev):
            rc = _lib.load().mpcrl_replay_sample(ptr(self.data), L, nx, self.E, self.cap, steps, ptr(idx), n,
                                                 ptr(self.pos_t if exclude_pos is None else exclude_pos), int(exclude_pos is not None),
                                                 ptr(self.iter_ok) if st else None, ptr(rows), ptr(x64[0]), ptr(x64[1]),
                                                 ptr(st[0]) if st else None, ptr(st[1]) if st else None, ptr(st[2]) if st else None,
                                                 ptr(st[3]) if st else None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"mpcrl_replay_sample failed with code {rc}")
        self.last_idx, self.last_steps, self.last_rows, self.last_x64, self.last_starts = idx, steps, rows, (x64[0], x64[1]), st
        return rows[:, :nx], rows[:, nx: 2 * nx], rows[:, 2 * nx: 2 * nx + nu], rows[:, -2], rows[:, -1]

    def iterates_of_last_sample(self):
        """For the transitions of the last sample(): (rows for obs, ok, rows for next_obs, ok) — row numbers into ``iters``.  The
        iterate for obs is the roll-out policy's own solution there.  For next_obs it is the roll-out solution of the NEXT step of the
        same environment where that exists and is that state (the episode went on, the slot is not the one about to be overwritten /
        not yet written) — else the solution at obs, one step earlier: the warm start the closed loop itself uses."""
        idx, steps, E = self.last_idx, self.last_steps, self.E
        step, env = idx // E, idx % E
        nstep = (step + 1) % self.cap
        flat = lambda t: t.reshape(self.cap * E, *t.shape[2:])
        cont = (self.last_rows[:, -1] == 0) & (nstep != self.pos_t) & (nstep < steps)
        nidx = torch.where(cont, nstep * E + env, idx)
        ok = flat(self.iter_ok)
        return idx, ok[idx], nidx, ok[nidx]


class BatchedTD3:
    """TD3 with the MPC as the actor, closed loop over E batched environments per rank (BASELINE config 5).

    What stable_baselines3's ``TD3`` does with ``MPCTD3Policy`` (rlmpc/td3/policies.py:225-361, driven by
    scripts/cartpole_mpc_as_td3_agent_closed_loop.py:40-67), with every per-observation loop turned into one batched solve:

      collect   a_t = clip(actor(s_t) + N(0, action_noise), -1, 1); env.step; replay.add; environments that ended are reset and their
                OCP instances start cold in the next solve (``mpc.reset(obs)``, script lines 62-64)
      train     sample B transitions; target action = clip(actor_target(s') + clip(N(0, target_noise), +-noise_clip), -1, 1) — the
                target actor is the same MPC with Polyak-averaged parameters theta'; y = r + gamma (1 - done) min_i Q_i'(s', a');
                critic loss = sum_i mse(Q_i(s, a), y); every ``policy_delay`` updates the deterministic policy gradient
                theta += lr_actor * mean_i (dpi/dtheta_i' grad_a Q_1(s_i, pi(s_i)))   (the step the reference leaves as a stub,
                "NOTE: The actor optimizer is not being used", policies.py:332), then Polyak updates of both targets.
      ranks     every rank owns its environments, its replay buffer and its solver handles; per update the critic gradients and the
                theta-gradient sums travel in ONE all-reduce (RCCL over xGMI with backend "nccl"), so all ranks hold identical
                critics and identical theta.
    The reference env's ``reward`` is the quadratic x^2 + theta^2 (continuous_cartpole/environment.py:193-194) — a cost; the default
    ``reward_scale = -1`` makes TD3 maximise its negative."""

    def __init__(self, ocp, env, batch_size: int = 256, buffer_steps: int = 64, gamma: float = 0.99, tau: float = 0.005,
                 policy_delay: int = 2, action_noise: float = 0.1, target_noise: float = 0.2, noise_clip: float = 0.5,
                 lr_critic: float = 1e-3, lr_actor: float = 1e-4, reward_scale: float = -1.0, net_arch=(64, 64), device=None,
                 group=None, seed: int = 0, learn_mask: Optional[torch.Tensor] = None, actor_factory=None, replay_iterates: bool = False,
                 blas: Optional[str] = "cublas", fused_critic: bool = True, pipeline: bool = False):
        """actor_factory(batch) -> an MPCActor-like object (default: MPCActor on the GPU).  The CPU tests of the loop's plumbing
        (replay, critic update, the single all-reduce) pass a closed-form stand-in policy; the product path never does.
        replay_iterates: keep, with every transition, the solver iterate the roll-out policy ended with (x, u, pi, bound multipliers and
        slacks: 9.9 KB per cartpole transition, float64) and start the two replay solves of an update from it — the target actor's at
        s' from the roll-out solution of the next step of that environment, the policy's at s from the roll-out solution at s —
        instead of from the cold iterate.  theta' and theta differ from the roll-out's theta by a few small policy steps, so the stored
        iterate is all but the answer: 0-4 SQP iterations instead of 6.6 on average and 22 for the slowest instance of a batch, which
        is what a lock-step launch lasts for.  (The reference's SB3 loop never resets its one solver between replay samples either:
        it warm-starts every solve from the previous — unrelated — sample's solution, rlmpc/td3/policies.py:186-213.)"""
        self.env, self.E, self.B = env, env.num_envs, batch_size
        if blas and torch.cuda.is_available() and actor_factory is None:
            # The critics' GEMMs are tiny (batch x 64 by 64 x 64): torch's default on ROCm sends nn.Linear through hipBLASLt, whose
            # kernels take 18 / 32 / 18 us for the forward / weight-gradient / input-gradient product at batch 4096 where rocBLAS
            # ("cublas" in torch's naming) takes 8 / 15 / 7 us — ~30 such products per step: 1.68 -> 1.61 ms (profiles/r06_td3_replay_iterates.txt).
            # A process-wide torch setting (blas=None leaves it alone).
            torch.backends.cuda.preferred_blas_library(blas)
        make = actor_factory or (lambda batch: MPCActor(ocp, batch, device))
        self.actor = make(self.E)                               # roll-out: keeps one warm-start iterate per environment
        dev = self.actor.mpc.device
        self.device = dev
        self.pi_mpc = make(batch_size)                          # pi(s_i) + dpi/dtheta_i on replay samples
        self.target_mpc = make(batch_size)                      # actor_target(s'_i): parameters theta'
        self.theta = self.actor.theta.clone()
        self.theta_target = self.theta.clone()
        # which entries of theta the policy gradient may move (default: the model block; W / yref have zero du0/dp anyway)
        self.learn_mask = torch.zeros_like(self.theta) if learn_mask is None else learn_mask.to(self.theta)
        if learn_mask is None:
            self.learn_mask[: ocp.n_model_p] = 1.0
        torch.manual_seed(seed)                                 # identical critic initialisation on every rank
        self.critic = ContinuousCritic(ocp.nx, ocp.nu, net_arch).to(dev)
        self.critic_target = ContinuousCritic(ocp.nx, ocp.nu, net_arch).to(dev)
        self.critic_target.load_state_dict(self.critic.state_dict())
        # fused / multi-tensor updates: the critic is 12 small tensors, one launch per tensor and operation is pure launch overhead
        # (capturable: the step counter lives on the device, so that the optimiser step can be part of a replayed HIP graph)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=lr_critic,
                                           **({"fused": True, "capturable": True} if dev.type == "cuda" else {}))
        self.replay_iterates = bool(replay_iterates)
        self._fused = actor_factory is None and dev.type == "cuda"      # the actors' output stage through mpcrl_policy_action (one launch)
        # the critic step through mpcrl_critic_td_grad / mpcrl_critic_dq_da (two launches + one, for ~60 + ~15 of the framework: 361 us of a
        # 1.33 ms step at batch 4096) where the critics have the shape those kernels are written for; any other net_arch stays on autograd
        self._fused_critic = fused_critic and self._fused and tuple(net_arch) == (64, 64) and ocp.nx + ocp.nu <= 64
        self._fused_sample = self._fused      # the replay batch through mpcrl_replay_sample (one launch after the draw)
        # pipeline (opt-in; step() with graphs on one rank): the update runs BESIDE the roll-out step of the same call instead of after
        # it — a pipelined actor / learner.  The update then samples the replay table without the slot the roll-out is writing (its
        # newest data are one step older) and the roll-out uses the policy parameters of before this call's update; the work per step
        # is the same.  What it buys: a lock-step solve lasts as long as its hardest instance, and the other solves fill the idle chip.
        self.pipeline = bool(pipeline)
        self._pipe_pos = None       # [1] int64 while a pipelined step is being captured: the slot the concurrent roll-out writes
        # the roll-out after the solve through mpcrl_td3_cartpole_collect (one launch for ~36) where the environment is the library's cartpole
        from .envs import BatchedCartPoleSwingUpEnv
        self._fused_collect = (self._fused and isinstance(env, BatchedCartPoleSwingUpEnv) and env._native() and env.dtype == torch.float64
                               and ocp.nu == 1 and ocp.nx == 4 and env.state.device == self.theta.device)
        if self._fused_critic:
            self._crit_flat, self._crit_grad = flatten_parameters(self.critic)
            self._crit_target_flat, _ = flatten_parameters(self.critic_target)
            self._crit_ws = self._dpg_ws = None
        nw_ = ocp.nx + ocp.nu
        self.buffer = DeviceReplayBuffer(buffer_steps, self.E, ocp.nx, ocp.nu, dev, iterate_dims=(
            (ocp.N + 1) * ocp.nx, ocp.N * ocp.nu, ocp.N * ocp.nx, 10 * (ocp.N + 1) * nw_) if self.replay_iterates else None)
        self.gamma, self.tau, self.policy_delay = gamma, tau, policy_delay
        self.action_noise, self.target_noise, self.noise_clip = action_noise, target_noise, noise_clip
        self.lr_actor, self.reward_scale, self.group = lr_actor, reward_scale, group
        rank = torch.distributed.get_rank(group) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        self.gen = torch.Generator(device=dev).manual_seed(seed + 1000 * rank)     # exploration / sampling differ per rank
        self.n_updates = 0
        self.obs = self.env.reset().to(dev)
        self._ended = None
        if self._fused_collect:
            self.obs = self.obs.to(torch.float64).contiguous()
            self._ended = torch.zeros(self.E, dtype=torch.int32, device=dev)      # updated in place: the cold mask of the next roll-out solve
            self._iter_rows = torch.zeros(self.E, dtype=torch.int64, device=dev) if self.replay_iterates else None
            self._collect_ws = torch.zeros(2 + 3 * ((self.E + 255) // 256), dtype=torch.float64, device=dev)
            self._lo_hi = (float(ocp.lbu[0]), float(ocp.ubu[0]))
        self._stats = torch.zeros(3, dtype=torch.float64, device=dev)    # reward sum, converged solves, episodes ended (since the last read)
        self._stats_steps = 0
        self._graphs = None
        self.n_crit = sum(p.numel() for p in self.critic.parameters())

    # ------------------------------------------------------------------ roll-out
    def _collect_step(self, static: bool = False):
        """One closed-loop step of all E environments.  static: the form that can be captured into a HIP graph and replayed — the
        replay write position is read on the device, the running observation / ended mask / statistics are updated in place."""
        if self._fused_collect:
            return self._collect_step_fused(static)
        r = self.actor.mpc.solve(self.obs.to(torch.float64), cold_mask=self._ended)   # ONE launch for E policies
        # a solve that ended with status 1 / 4 may hand back a non-finite u0: such an environment gets the zero action (a
        # finite fallback; the reference raises instead, mpc.py:81-83).  The stored transition is the one that really happened
        # (action 0 was applied), so no NaN ever reaches the environment, the replay buffer or the critic
        eps = torch.randn(r.u0.shape, dtype=torch.float32, device=self.device, generator=self.gen)
        if self._fused:
            a, u_ok = self.actor.action(r, eps, sigma=self.action_noise, accept_status2=True)
        else:
            u_ok = torch.isfinite(r.u0).all(dim=1) & ((r.status == 0) | (r.status == 2))
            a = torch.where(u_ok[:, None], self.actor.scale_action(torch.nan_to_num(r.u0)), torch.zeros_like(r.u0)).to(torch.float32)
            a = (a + self.action_noise * eps).clamp(-1.0, 1.0)
        nxt, rew, term, trunc = self.env.step(a.to(self.env.device))
        nxt, rew = nxt.to(self.device), rew.to(self.device)
        done = (term | trunc).to(self.device)
        itk = {}
        if self.replay_iterates:       # the iterate this solve ended with, next to its transition
            itk = {"iterate": self.actor.mpc, "iterate_ok": u_ok & (r.status == 0)}
        if static:
            self.buffer.add_at_device_pos(self.obs, nxt, a, self.reward_scale * rew, term.to(self.device), **itk)
        else:
            self.buffer.add(self.obs, nxt, a, self.reward_scale * rew, term.to(self.device), **itk)
        self._stats[0] += rew.sum()
        self._stats[1] += (r.status == 0).sum()
        self._stats[2] += done.sum()
        if hasattr(self.env, "reset_where"):
            # no host synchronisation in a step: the masked reset and the per-instance cold mask run every step (empty masks are no-ops)
            new_obs = self.env.reset_where(done.to(self.env.device)).to(self.device)
            if static:
                self.obs.copy_(new_obs)
                self._ended.copy_(done)
            else:
                self.obs, self._ended = new_obs, done
        else:
            any_done = bool(done.any())      # an episode end changes the control flow: one host synchronisation per step
            self.obs = self.env.reset(done.to(self.env.device)).to(self.device) if any_done else nxt
            self._ended = done if any_done else None

    def _collect_step_fused(self, static: bool) -> None:
        """_collect_step on mpcrl_td3_cartpole_collect: the solve, the two draws (the generators stay torch's), ONE launch for the actor's
        output stage, the environment step, the replay row, the statistics, the resets, the next observation and cold mask — all in
        place, so eager and graph-replayed steps are the same launches — and the move of the iterates into the replay tables."""
        from . import _lib
        from .batch import _ptr
        env, buf, dev = self.env, self.buffer, self.device
        r = self.actor.mpc.solve(self.obs, cold_mask=self._ended)
        eps = torch.randn(r.u0.shape, dtype=torch.float32, device=dev, generator=self.gen)
        u01 = torch.rand(self.E, generator=env.gen, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().mpcrl_td3_cartpole_collect(
                env._par(), self.E, _ptr(env.state), _ptr(env.steps), _ptr(r.u0), _ptr(r.status), _ptr(eps), _ptr(u01), self._lo_hi[0],
                self._lo_hi[1], int(bool(self.actor.scale)), float(self.action_noise), _ptr(self.obs), _ptr(self._ended), _ptr(buf.data),
                buf.cap, float(self.reward_scale), _ptr(buf.pos_t), _ptr(buf.iter_ok) if self.replay_iterates else None,
                _ptr(self._iter_rows) if self.replay_iterates else None, _ptr(self._stats), _ptr(self._collect_ws),
                torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mpcrl_td3_cartpole_collect failed with code {rc}")
        if self.replay_iterates:
            self.actor.mpc.get_iterate_rows(*buf.iters, index=self._iter_rows)
        if not static:
            buf._advance()

    def collect(self, n_steps: int, stats: bool = True) -> dict:
        """n_steps closed-loop steps of all E environments; returns roll-out statistics (device tensors -> python floats once;
        stats=False: no host synchronisation at all, read them later with last_stats())."""
        if stats:
            self._stats.zero_()
            self._stats_steps = 0
        for _ in range(n_steps):
            if self._graphs is not None:
                self._graphs["collect"].replay()
                self.buffer._advance()
            else:
                self._collect_step()
        self._stats_steps += n_steps
        return self.last_stats() if stats else {}

    def last_stats(self) -> dict:
        n = max(1, self._stats_steps) * self.E
        v = self._stats.tolist()
        return {"mean_reward": v[0] / n, "converged_fraction": v[1] / n, "episodes_ended": int(v[2])}

    # ------------------------------------------------------------------ learning
    def _allreduce(self, flat: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _update_pre(self, do_policy: bool):
        """Everything of one update up to the message of the collective: sample, target actor's batched solve, critic loss and
        gradients, (every policy_delay-th update) the policy's batched solve with du0*/dtheta and the theta-gradient sum.
        Returns (flat message, loss)."""
        world, n_theta = self._world(), self.theta.numel()
        if self._fused_sample:      # the draw, then ONE launch for the gather, the index arithmetic, the masks and the float64 states
            obs, nxt, act, rew, done = self.buffer.sample_fused(self.B, self.gen, exclude_pos=self._pipe_pos)
            obs64, nxt64 = self.buffer.last_x64
        else:
            obs, nxt, act, rew, done = self.buffer.sample(self.B, self.gen)
            obs64 = nxt64 = None
        with torch.no_grad():
            eps = torch.randn(act.shape, dtype=torch.float32, device=self.device, generator=self.gen)
            if self.replay_iterates:
                if self._fused_sample:
                    it_s, cold_s, it_n, cold_n = self.buffer.last_starts
                else:
                    it_s, ok_s, it_n, ok_n = self.buffer.iterates_of_last_sample()
                    cold_s, cold_n = ~ok_s, ~ok_n
                self._load_iterate(self.target_mpc.mpc, it_n)
                # (a stored iterate of a failed roll-out solve is not a starting point: that instance starts cold)
                # (no packing order: nearly every instance is converged at its start, there is no difficulty to group by — 14 us of order kernel)
                rt = self.target_mpc.mpc.solve(nxt.to(torch.float64) if nxt64 is None else nxt64, cold_mask=cold_n, reorder=False)     # actor_target(s'), one launch, warm
            else:
                rt = self.target_mpc.mpc.solve(nxt.to(torch.float64) if nxt64 is None else nxt64, cold=True)           # actor_target(s'), one launch
            # failed target solves are SELECTED out (a product with a 0 / 1 mask would keep their NaN: NaN * 0 = NaN), and so are
            # transitions whose stored observations are not finite
            # (one finiteness test over the whole transition: every separate test, fill and select is a launch of its own)
            if self._fused:      # the target actor's output stage in one launch; the sampled rows are one packed tensor already
                a_next, ok_u = self.target_mpc.action(rt, eps, sigma=self.target_noise, noise_clip=self.noise_clip)
                if not self._fused_critic:
                    row = self.buffer.last_rows
                    ok_b = ok_u & torch.isfinite(row).all(dim=1)
            else:
                noise = (self.target_noise * eps).clamp(-self.noise_clip, self.noise_clip)
                row = torch.cat([obs, nxt, act, rew[:, None], rt.u0.to(obs.dtype)], dim=1)
                ok_b = (rt.status == 0) & torch.isfinite(row).all(dim=1)
                u_next = torch.where(ok_b[:, None], rt.u0, 0.0)
                a_next = (self.target_mpc.scale_action(u_next).to(torch.float32) + noise).clamp(-1.0, 1.0)
            nx_ = obs.shape[1]
            # one flat message: critic gradients (already the local mean) | theta-gradient sum | sample count
            flat = torch.zeros(self.n_crit + n_theta + 1, dtype=torch.float64, device=self.device)
            if self._fused_critic and not self._fused:
                raise RuntimeError("BatchedTD3: fused_critic needs the fused actor output stage (_fused)")
            if self._fused_critic:      # TD target, both critics' loss and its gradient into the message: two launches
                loss, _, self._crit_ws = critic_td_grad(self.buffer.last_rows, nx_, act.shape[1], a_next, ok_u, self._crit_flat, self._crit_target_flat,
                                                        len(self.critic.q_networks), self.gamma, 1.0 / world, flat, self._crit_ws)
                loss = loss[0]
        if not self._fused_critic:
            with torch.no_grad():
                safe = torch.where(ok_b[:, None], row, 0.0)
                obs_s, nxt_s, act_s = safe[:, :nx_], safe[:, nx_: 2 * nx_], safe[:, 2 * nx_: 2 * nx_ + act.shape[1]]
                q_next = torch.min(*self.critic_target(nxt_s, a_next)).squeeze(1)
                ok_t = ok_b.to(torch.float32)
                y = torch.where(ok_b, rew + self.gamma * (1.0 - done) * q_next, 0.0)
            qs = self.critic(obs_s, act_s)
            loss = sum((torch.where(ok_b, q.squeeze(1) - y, 0.0) ** 2).sum() for q in qs) / ok_t.sum().clamp(min=1.0)
            self.critic_opt.zero_grad(set_to_none=True)
            loss.backward()
            flat[: self.n_crit] = torch.cat([p.grad.reshape(-1) for p in self.critic.parameters()]).to(torch.float64) / world
        if do_policy:
            if self.replay_iterates:
                self._load_iterate(self.pi_mpc.mpc, it_s)
                rp = self.pi_mpc.mpc.solve(obs.to(torch.float64) if obs64 is None else obs64, sens_pi=True, cold_mask=cold_s, reorder=False)   # pi(s_i), dpi/dtheta_i: one launch, warm
            else:
                rp = self.pi_mpc.mpc.solve(obs.to(torch.float64) if obs64 is None else obs64, sens_pi=True, cold=True)   # pi(s_i), dpi/dtheta_i: one launch
            ocp_nu = act.shape[1]
            if self._fused_critic:
                a_pi, ok_u = self.pi_mpc.action(rp)
                dq_da, okb = critic_dq_da(self.buffer.last_rows, nx_, a_pi, ok_u, self._crit_flat)     # dQ_1/da at (s, pi(s)): one launch
            elif self._fused:
                a_pi, ok_u = self.pi_mpc.action(rp)
                okb = ok_u & torch.isfinite(obs).all(dim=1)
                a_pi = a_pi.requires_grad_(True)
            else:
                okb = (rp.status == 0) & torch.isfinite(torch.cat([rp.u0.to(obs.dtype), obs], dim=1)).all(dim=1)
                u_pi = torch.where(okb[:, None], rp.u0, 0.0)
                a_pi = self.pi_mpc.scale_action(u_pi).to(torch.float32).detach().requires_grad_(True)
            # the policy branch has its own mask: obs_s above is zeroed where the TARGET solve failed, and dQ/da at obs = 0 paired
            # with pi(s) and dpi/dtheta of the real obs would bias the step for rows whose policy solve succeeded
            if not self._fused_critic:
                obs_p = torch.where(okb[:, None], obs, 0.0)
                (dq_da,) = torch.autograd.grad(self.critic.q1_forward(obs_p, a_pi).sum(), a_pi)
            if self._fused_critic and ocp_nu <= 8:      # the contraction with dpi/dtheta and the count into the message: one launch
                self._dpg_ws = dpg_grad(dq_da, okb, rp.dpi_dp, self.pi_mpc.low, self.pi_mpc.high, self.pi_mpc.scale, flat[self.n_crit:], self._dpg_ws)
            else:
                chain = (2.0 / (self.pi_mpc.high - self.pi_mpc.low)) if self.pi_mpc.scale else torch.ones_like(self.pi_mpc.low)
                okp = okb.to(torch.float64)
                g = torch.einsum("bu,bup->bp", torch.where(okb[:, None], dq_da.to(torch.float64) * chain, 0.0), torch.nan_to_num(rp.dpi_dp))
                flat[self.n_crit: self.n_crit + n_theta] = g.sum(0)
                flat[-1] = okp.sum()
        return flat, loss.detach()

    def _load_iterate(self, mpc, rows) -> None:
        mpc.set_iterate_rows(*self.buffer.iters, index=rows.contiguous())       # ONE launch: stored iterate i := table row rows[i]

    def _push_theta(self) -> None:
        for m, th in ((self.actor, self.theta), (self.pi_mpc, self.theta), (self.target_mpc, self.theta_target)):
            m.theta = th
            m.mpc.set_theta(th)

    def _update_post(self, flat: torch.Tensor, do_policy: bool, push_theta: bool = True):
        """After the collective: the averaged critic gradients into the optimiser step; the policy step and the Polyak updates.
        All parameter tensors are updated IN PLACE (their addresses are what a captured graph and the solver handles hold)."""
        n_theta = self.theta.numel()
        params = list(self.critic.parameters())
        if self._fused_critic:
            self._crit_grad.copy_(flat[: self.n_crit])        # every p.grad is a view of it: one launch
        else:
            g32, off, views = flat[: self.n_crit].to(params[0].dtype), 0, []
            for p in params:
                views.append(g32[off: off + p.numel()].view(p.shape))
                off += p.numel()
            torch._foreach_copy_([p.grad for p in params], views)
        self.critic_opt.step()
        step = None
        if do_policy and self._fused_critic:      # the policy step, theta' and the target critics' Polyak update: one launch
            from . import _lib
            from .batch import _ptr
            step = torch.empty_like(self.theta)
            with torch.cuda.device(self.device), torch.no_grad():
                rc = _lib.load().mpcrl_td3_policy_post(C.c_void_p(flat.data_ptr() + 8 * self.n_crit), n_theta, float(self.lr_actor), _ptr(self.learn_mask),
                                                       float(self.tau), _ptr(self.theta), _ptr(self.theta_target), _ptr(step), _ptr(self._crit_flat),
                                                       _ptr(self._crit_target_flat), self.n_crit, torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"mpcrl_td3_policy_post failed with code {rc}")
            if push_theta:
                self._push_theta()
        elif do_policy:
            step = self.lr_actor * self.learn_mask * flat[self.n_crit: self.n_crit + n_theta] / flat[-1].clamp(min=1.0)
            self.theta.add_(step)
            self.theta_target.mul_(1.0 - self.tau).add_(self.theta, alpha=self.tau)
            if push_theta:
                self._push_theta()
            with torch.no_grad():
                if self._fused_critic:
                    self._crit_target_flat.mul_(1.0 - self.tau).add_(self._crit_flat, alpha=self.tau)
                else:
                    pts = list(self.critic_target.parameters())
                    torch._foreach_mul_(pts, 1.0 - self.tau)
                    torch._foreach_add_(pts, params, alpha=self.tau)
        return step

    def train(self, n_updates: int, stats: bool = True) -> dict:
        loss, step = None, None
        for _ in range(n_updates):
            self.n_updates += 1
            do_policy = self.n_updates % self.policy_delay == 0
            if self._graphs is not None:
                gp = self._graphs["update"][do_policy]
                gp["pre"].replay()
                self._allreduce(gp["flat"])
                gp["post"].replay()
                loss, step = gp["loss"], gp["step"]
            else:
                flat, loss = self._update_pre(do_policy)
                flat = self._allreduce(flat)
                step = self._update_post(flat, do_policy)
            self._last_loss = loss
        if not stats:
            return {}
        return {"critic_loss": float(loss.item()) if loss is not None else 0.0,
                "theta_step_norm": float(step.norm().item()) if step is not None else 0.0}

    def step(self, n_steps: int = 1) -> None:
        """n closed-loop steps, each collect(1) + train(1) without statistics (read them with last_stats() / last_critic_loss()).  With
        graphs on a single rank a step is ONE graph replay."""
        for _ in range(n_steps):
            if self._graphs is not None and "full" in self._graphs:
                self.n_updates += 1
                gp = self._graphs["full"][self.n_updates % self.policy_delay == 0]
                gp["graph"].replay()
                self.buffer._advance()
                self._stats_steps += 1
                self._last_loss = gp["loss"]
            else:
                self.collect(1, stats=False)
                self.train(1, stats=False)

    def last_critic_loss(self) -> float:
        return float(self._last_loss.item()) if getattr(self, "_last_loss", None) is not None else float("nan")

    def _buffer_tensors(self):
        b = self.buffer
        return [b.data] + ((b.iters + [b.iter_ok]) if b.iters is not None else [])

    # ------------------------------------------------------------------ HIP graphs
    def enable_graphs(self) -> None:
        """Captures the roll-out step and the two halves of an update (before / after the collective; with and without the policy
        step) into HIP graphs and replays them from then on: a closed-loop step is ~190 small launches of which the three batched
        solves are the only ones that take time — replayed as four graph launches the step is no longer bound by the host's launch
        rate.  Every library call is capture-safe (no synchronisation, no allocation, fixed launch sequence: include/mpcrl.h).
        Requirements: CUDA device, a native batched environment (reset_where), the replay buffer full (its sampling range is then a
        constant; the missing steps are collected here).  The collective stays an eager call between the two halves."""
        if self._graphs is not None:
            return
        if self.device.type != "cuda" or not hasattr(self.env, "reset_where"):
            raise RuntimeError("enable_graphs needs a CUDA device and an environment with reset_where")
        if hasattr(self.env, "_native") and not self.env._native():
            # a replayed step must update the environment's own state tensors in place (the library's environment kernels do)
            raise RuntimeError("enable_graphs needs the environment on the GPU (its state is updated in place by the library kernels)")
        while not self.buffer.full:
            self._collect_step()
        if self._ended is None:
            self._ended = torch.zeros(self.E, dtype=torch.bool, device=self.device)
        self.buffer.pos_t.fill_(self.buffer.pos)
        # The replay solves' launch shape is chosen by timing (mpcrl_set_launch_mode) and a captured graph keeps the shape preferred
        # at capture: let the two handles finish their probes on stored replay rows first (no random draws, no agent state touched).
        rows = self.buffer.next_obs.reshape(-1, self.buffer.next_obs.shape[-1])
        rows = rows[torch.arange(self.B, device=self.device) * (rows.shape[0] // self.B if rows.shape[0] >= self.B else 1) % rows.shape[0]]
        rows = torch.where(torch.isfinite(rows), rows, torch.zeros_like(rows)).to(torch.float64).contiguous()
        if self.replay_iterates:
            # warm replay solves: a handful of SQP rounds for a few instances, none for most — the plain launch shape (three instances per
            # wavefront advance together, nobody parked) measured the same or better than the time-sliced one on such batches
            # (0.24 vs 0.29 ms per 4096, profiles/r06_td3_replay_iterates.txt); no probe calls, the shape is fixed
            self.target_mpc.mpc.set_launch_mode(-1)
            self.pi_mpc.mpc.set_launch_mode(-1)
            # the roll-out too: its launch lasts as long as the few just-reset (cold) instances, and the time-sliced shape parks every
            # instance a quarter of the time (three A/B pairs of the whole step: 1.502 / 1.510 / 1.507 -> 1.478 / 1.482 / 1.478 ms)
            self.actor.mpc.set_launch_mode(-1)
        for _ in range(0 if self.replay_iterates else 4):
            self.target_mpc.mpc.solve(rows, cold=True)
            self.pi_mpc.mpc.solve(rows, sens_pi=True, cold=True)
            torch.cuda.synchronize(self.device)
        gens = [self.gen] + ([self.env.gen] if hasattr(self.env, "gen") else [])
        side = torch.cuda.Stream(device=self.device)
        # The warm-up on the capture stream (allocator, lazy initialisations) runs both kinds of update and one roll-out step for real.
        # Everything it touches is put back afterwards — critics, optimiser state, theta / theta', generators, environment, replay
        # buffer, running observation, statistics — so that enabling graphs neither perturbs the training trajectory nor shifts the
        # policy_delay phase relative to eager mode.
        import copy
        snap = {"critic": copy.deepcopy(self.critic.state_dict()), "critic_target": copy.deepcopy(self.critic_target.state_dict()),
                "opt": copy.deepcopy(self.critic_opt.state_dict()), "theta": self.theta.clone(), "theta_target": self.theta_target.clone(),
                "gens": [g.get_state() for g in gens], "obs": self.obs.clone(), "ended": self._ended.clone(), "stats": self._stats.clone(),
                "env": (self.env.state.clone(), self.env.steps.clone()) if hasattr(self.env, "steps") else None,
                "buf": [t.clone() for t in self._buffer_tensors()],
                "pos": (self.buffer.pos, self.buffer.full), "iter": self.actor.mpc.get_iterate() if hasattr(self.actor.mpc, "get_iterate") else None,
                "iter_flags": (self.actor.mpc.has_iterate, self.actor.mpc.duals_valid)}
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for dp in (False, True):
                flat, _ = self._update_pre(dp)
                self._update_post(self._allreduce(flat), dp)
            self._collect_step(static=True)
            self.buffer._advance()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            self.critic.load_state_dict(snap["critic"]), self.critic_target.load_state_dict(snap["critic_target"])
            # in place: the optimiser's state tensors (device step counter included) are what a captured update replays on
            st_now, st_old = self.critic_opt.state_dict()["state"], snap["opt"]["state"]
            for k, d in st_now.items():
                for name, v in d.items():
                    if torch.is_tensor(v):
                        v.copy_(st_old[k][name]) if k in st_old else v.zero_()
            self.theta.copy_(snap["theta"]), self.theta_target.copy_(snap["theta_target"])
            for m, th in ((self.actor, self.theta), (self.pi_mpc, self.theta), (self.target_mpc, self.theta_target)):
                m.mpc.set_theta(th)
            for g, stt in zip(gens, snap["gens"]):
                g.set_state(stt)
            self.obs.copy_(snap["obs"]), self._ended.copy_(snap["ended"]), self._stats.copy_(snap["stats"])
            if snap["env"] is not None:
                self.env.state.copy_(snap["env"][0]), self.env.steps.copy_(snap["env"][1])
            for t, old in zip(self._buffer_tensors(), snap["buf"]):
                t.copy_(old)
            self.buffer.pos, self.buffer.full = snap["pos"]
            self.buffer.pos_t.fill_(self.buffer.pos)
            if snap["iter"] is not None:
                # the roll-out actor's warm start exactly as it was: iterate, and whether its bound multipliers came from a solve
                # (the replay handles only ever solve cold: their stored iterates are never read)
                x_, u_, pi_, bnd_, _ = snap["iter"]
                had, duals = snap["iter_flags"]
                if not had:
                    self.actor.mpc.reset()
                else:
                    self.actor.mpc.set_iterate(x_, u_, pi_, bnd_ if duals else None)
        torch.cuda.synchronize(self.device)

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            for gen in gens:
                g.register_generator_state(gen)
            with torch.cuda.graph(g, stream=side):
                out = fn()
            return g, out

        graphs = {"update": {}}
        for dp in (False, True):
            g_pre, (flat, loss) = capture(lambda: self._update_pre(dp))
            g_post, step = capture(lambda: self._update_post(flat, dp))
            graphs["update"][dp] = {"pre": g_pre, "post": g_post, "flat": flat, "loss": loss, "step": step}
        graphs["collect"], _ = capture(lambda: self._collect_step(static=True))
        if self._world() == 1:
            # one rank: nothing has to happen between the halves of an update, so a whole closed-loop step — roll-out, update before and
            # after the (absent) collective — is ONE graph per kind of update (step()): two graph boundaries and the generator-state
            # fills of two replays less per step
            pipe = self.pipeline and self._fused_collect and self._fused_sample
            if self.pipeline and not pipe:
                raise RuntimeError("BatchedTD3(pipeline=True) needs the library's cartpole environment on the GPU (the fused roll-out and replay kernels)")
            if pipe:
                self._pipe_pos_t = torch.zeros(1, dtype=torch.int64, device=self.device)
                self._pipe_stream = torch.cuda.Stream(device=self.device)

            def whole(dp):
                if not pipe:
                    self._collect_step(static=True)
                    flat, loss = self._update_pre(dp)
                    return loss, self._update_post(flat, dp)
                # pipelined: [update on the table without the slot being written] beside [roll-out step]; the handles take the new theta
                # after both (the roll-out solve reads its handle's parameters while it runs)
                self._pipe_pos_t.copy_(self.buffer.pos_t)
                cur = torch.cuda.current_stream(self.device)
                self._pipe_stream.wait_stream(cur)
                with torch.cuda.stream(self._pipe_stream):
                    self._pipe_pos = self._pipe_pos_t
                    flat, loss = self._update_pre(dp)
                    self._pipe_pos = None
                    out = loss, self._update_post(flat, dp, push_theta=False)
                self._collect_step(static=True)
                cur.wait_stream(self._pipe_stream)
                if dp:
                    self._push_theta()
                return out
            graphs["full"] = {}
            for dp in (False, True):
                g, (loss, step) = capture(lambda: whole(dp))
                graphs["full"][dp] = {"graph": g, "loss": loss, "step": step}
        torch.cuda.synchronize(self.device)
        self._graphs = graphs
