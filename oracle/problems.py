"""Problem data of the reference's three OCPs, restated in torch float64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every number here is taken from
the reference file cited next to it; derivatives are left to torch autograd so
that this file is an independent check of the engine's hand-written ones.

A ``Problem`` describes  (notation of rlmpc/mpc/nlp.py:884-1275)

    min  sum_{k<N} c_k l_k(x_k,u_k,p) + c_N l_N(x_N,p) + slack penalties
    s.t. x_{k+1} = F(x_k,u_k,p),   x_0 = x0  (two inequalities in the mirror)
         lbu <= u_k <= ubu                     k = 0..N-1
         lbx <= x_k[idxbx] <= ubx  (soft on idxsbx)   k = 1..N-1
         lbx_e <= x_N[idxbx_e] <= ubx_e

with ``p`` the *full* parameter vector in the reference's order
(nlp.py:969-989: ``model`` block first, then W_0, W, W_e, yref_0, yref, yref_e,
each flattened column-major).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

DT = torch.float64   # every tensor below is created from float64 numpy data or with an explicit dtype (no global default is touched)


@dataclass
class Problem:
    name: str
    nx: int
    nu: int
    N: int
    dT: float                       # tf / N          (nlp.py:1163-1164)
    p0: np.ndarray                  # nominal full parameter vector
    p_labels: List[str]
    F: Callable                     # F(x,u,p) -> x_next          (torch)
    stage_cost: Callable            # l(k,x,u,p) -> scalar, k<N   (torch, unscaled)
    terminal_cost: Callable         # l_e(x,p) -> scalar          (torch, unscaled)
    cost_kind: str                  # "NLS" (no gamma, nlp.py:1038-1055) or "EXTERNAL" (nlp.py:1078-1091)
    lbu: np.ndarray
    ubu: np.ndarray
    idxbx: np.ndarray = field(default_factory=lambda: np.zeros(0, int))
    lbx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0, int))
    lbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxsbx: np.ndarray = field(default_factory=lambda: np.zeros(0, int))  # positions inside idxbx
    zl: np.ndarray = field(default_factory=lambda: np.zeros(0))
    zu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    gamma: float = 1.0
    tol: float = 1e-6               # acados default nlp tol; chain 1e-5 (ocp_utils.py:311-312)
    max_iter: int = 500
    x0_default: Optional[np.ndarray] = None
    extra: dict = field(default_factory=dict)

    @property
    def n_p(self) -> int:
        return int(self.p0.shape[0])

    def cost_scaling(self, gamma: Optional[float] = None) -> np.ndarray:
        """c_k, k = 0..N.  NLS mirror: dT for k<N, 1 at N (nlp.py:1044-1055).
        EXTERNAL: dT at 0, gamma^k dT inside, gamma^N at N (nlp.py:1083-1091)."""
        g = self.gamma if gamma is None else gamma
        c = np.empty(self.N + 1)
        if self.cost_kind == "NLS":
            c[: self.N] = self.dT
            c[self.N] = 1.0
        else:
            c[0] = self.dT
            for k in range(1, self.N):
                c[k] = g ** k * self.dT
            c[self.N] = g ** self.N
        return c

    def slack_scaling(self, gamma: Optional[float] = None) -> np.ndarray:
        """weight on z^T s at stage k: dT * gamma^k for k = 1..N-1 (nlp.py:1118-1130)."""
        g = self.gamma if gamma is None else gamma
        return np.array([self.dT * g ** k for k in range(self.N + 1)])


# --------------------------------------------------------------------------------------
# integrators (rlmpc/common/integrator.py:6-33, chain_mass/ocp_utils.py:42-56)
# --------------------------------------------------------------------------------------
def rk4(f, x, u, p, h, n_steps=1):
    for _ in range(n_steps):
        k1 = f(x, u, p)
        k2 = f(x + h / 2 * k1, u, p)
        k3 = f(x + h / 2 * k2, u, p)
        k4 = f(x + h * k3, u, p)
        x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


# --------------------------------------------------------------------------------------
# cartpole  (rlmpc/mpc/cartpole/acados.py:28-108, config/cartpole.yaml)
# --------------------------------------------------------------------------------------
CARTPOLE_G = 9.8  # config/cartpole.yaml:75-78 (fixed)


def cartpole_ode(x, u, p):
    # acados.py:71-82, restated as written (no `l` factor in temp / x_ddot)
    M, m, l = p[0], p[1], p[2]
    s_dot, th, th_dot = x[1], x[2], x[3]
    c, s = torch.cos(th), torch.sin(th)
    temp = (u[0] + m * th_dot ** 2 * s) / (m + M)
    th_dd = (CARTPOLE_G * s - c * temp) / (l * (4.0 / 3.0 - m * c ** 2 / (m + M)))
    return torch.stack([s_dot, temp - m * th_dd * c / (m + M), th_dot, th_dd])


def make_cartpole(N: int = 20, tf: float = 2.0) -> Problem:
    """N=20, tf=2.0 keeps the reference's dt = tf/N = 0.1 (config/cartpole.yaml:7,23 has N=30, tf=3.0).
    RK4 sub-step h = tf/N/sim_method_num_stages = 0.025, ONE step per stage (acados.py:86-92)."""
    nx, nu = 4, 1
    h = tf / N / 4
    W = np.diag([10.0, 0.1, 10.0, 0.1, 0.01])          # yaml:29-42
    W_e = np.diag([10.0, 0.1, 10.0, 0.1])              # yaml:43-49
    yref = np.zeros(5)
    yref_e = np.zeros(4)
    p0 = np.concatenate([[1.0, 0.1, 0.5], W.flatten("F"), W.flatten("F"), W_e.flatten("F"), yref, yref, yref_e])
    labels = ["M", "m", "l"] + [f"W_0_{i}" for i in range(25)] + [f"W_{i}" for i in range(25)] + \
        [f"W_e_{i}" for i in range(16)] + [f"yref_0_{i}" for i in range(5)] + [f"yref_{i}" for i in range(5)] + \
        [f"yref_e_{i}" for i in range(4)]

    def F(x, u, p):
        return rk4(cartpole_ode, x, u, p, h, 1)

    def stage_cost(k, x, u, p):
        # non-parameterised NLS mirror (nlp.py:517-534,1039-1055): W, yref enter as NUMBERS — the values set_parameter / cost_set
        # put into the cost block of p (mpc.py:233-257), detached so that no derivative with respect to them exists
        pd = p.detach()
        Wk = (pd[3:28] if k == 0 else pd[28:53]).reshape(5, 5).T          # column-major blocks W_0 / W
        y = torch.cat([x, u]) - (pd[69:74] if k == 0 else pd[74:79])
        return 0.5 * y @ (Wk @ y)

    def terminal_cost(x, p):
        pd = p.detach()
        y = x - pd[79:83]
        return 0.5 * y @ (pd[53:69].reshape(4, 4).T @ y)

    xb = np.array([2.4, 10.0, 6.28, 10.0])             # yaml:86-91
    return Problem(
        name="cartpole", nx=nx, nu=nu, N=N, dT=tf / N, p0=p0, p_labels=labels,
        F=F, stage_cost=stage_cost, terminal_cost=terminal_cost, cost_kind="NLS",
        lbu=np.array([-30.0]), ubu=np.array([30.0]),
        idxbx=np.arange(4), lbx=-xb, ubx=xb, idxbx_e=np.arange(4), lbx_e=-xb, ubx_e=xb,
        tol=1e-6, max_iter=500, x0_default=np.array([0.0, 0.0, 3.14, 0.0]),
        extra={"h": h, "W": W, "W_e": W_e, "skip_corrector": True},
    )


# --------------------------------------------------------------------------------------
# linear system (rlmpc/mpc/linear_system/acados.py:27-131, tests/test_linear_example.py:9-17)
# --------------------------------------------------------------------------------------
def make_linear_system(gamma: float = 0.99, N: int = 40) -> Problem:
    from scipy.linalg import solve_discrete_are

    A = np.array([[1.0, 0.25], [0.0, 1.0]])
    B = np.array([[0.03125], [0.25]])
    P = solve_discrete_are(A, B, np.eye(2), np.eye(1))   # numeric, non-parametric (acados.py:51-57)
    Pt = torch.tensor(P)
    p0 = np.concatenate([A.flatten("F"), B.flatten("F"), [0.0, 0.0], [1e-3], [0.0, 0.0, 0.0]])
    labels = ["A_0", "A_1", "A_2", "A_3", "B_0", "B_1", "b_0", "b_1", "V_0", "f_0", "f_1", "f_2"]

    def F(x, u, p):
        Am = torch.stack([torch.stack([p[0], p[2]]), torch.stack([p[1], p[3]])])  # column-major reshape (acados.py:60-62)
        return Am @ x + p[4:6] * u[0] + p[6:8]

    def ext(x, u, p):
        y = torch.cat([x, u])
        return 0.5 * y @ y + p[9:12] @ y

    def stage_cost(k, x, u, p):
        return ext(x, u, p) + (p[8] if k == 0 else 0.0)   # l_0 = V_0 + l (acados.py:43-47)

    def terminal_cost(x, p):
        return 0.5 * x @ (Pt @ x)

    return Problem(
        name="linear_system", nx=2, nu=1, N=N, dT=1.0, p0=p0, p_labels=labels,
        F=F, stage_cost=stage_cost, terminal_cost=terminal_cost, cost_kind="EXTERNAL",
        lbu=np.array([-1.0]), ubu=np.array([1.0]),
        idxbx=np.arange(2), lbx=np.array([0.0, -1.0]), ubx=np.array([1.0, 1.0]),
        idxsbx=np.array([0]), zl=np.array([1e2]), zu=np.array([1e2]),
        gamma=gamma, tol=1e-6, max_iter=100, x0_default=np.array([0.5, 0.5]),
        extra={"P": P, "lq": True},
    )


# --------------------------------------------------------------------------------------
# chain of masses (rlmpc/mpc/chain_mass/ocp_utils.py:42-376, examples/chain_mass.py:17-25)
# --------------------------------------------------------------------------------------
def chain_layout(n_mass: int):
    M = n_mass - 2
    nl = n_mass - 1
    nx, nu = (2 * M + 1) * 3, 3
    off = {}
    o = 0
    for key, sz in [("m", nl), ("D", 3 * nl), ("L", 3 * nl), ("C", 3 * nl), ("Q", nx * nx), ("R", nu * nu), ("w", 3 * M)]:
        off[key] = (o, o + sz)
        o += sz
    return M, nl, nx, nu, off, o


def make_chain_ode(n_mass: int):
    M, nl, nx, nu, off, n_p = chain_layout(n_mass)

    def ode(x, u, p):
        pos = x[: 3 * (M + 1)].reshape(M + 1, 3)
        vel = x[3 * (M + 1):].reshape(M, 3)
        m = p[off["m"][0]: off["m"][1]]
        D = p[off["D"][0]: off["D"][1]].reshape(nl, 3)
        L = p[off["L"][0]: off["L"][1]].reshape(nl, 3)
        C = p[off["C"][0]: off["C"][1]].reshape(nl, 3)
        w = p[off["w"][0]: off["w"][1]].reshape(M, 3)
        z3 = torch.zeros(1, 3, dtype=x.dtype)
        dist = pos - torch.cat([z3, pos[:-1]])                                   # ocp_utils.py:80-84
        nrm = torch.sqrt((dist * dist).sum(dim=1, keepdim=True))
        Fs = D / m[:, None] * (1.0 - L / nrm) * dist                             # ocp_utils.py:86-88
        dv = torch.cat([vel, u[None, :]]) - torch.cat([z3, vel])                 # ocp_utils.py:99-105
        Ft = Fs + C * dv                                                         # ocp_utils.py:107-109
        grav = torch.tensor([0.0, 0.0, -9.81], dtype=x.dtype)
        f = -Ft[:M] + Ft[1:] + grav[None, :] + w                                  # ocp_utils.py:76-77,91-96,111-125
        return torch.cat([vel.reshape(-1), u, f.reshape(-1)])                    # ocp_utils.py:130

    return ode


def chain_steady_state(n_mass: int, p: np.ndarray, x_end: np.ndarray) -> np.ndarray:
    """Root of f_expl = 0 with the last mass pinned at x_end and u = 0
    (compute_parametric_steady_state, ocp_utils.py:150-192; the reference abuses IPOPT for it,
    here a plain Newton iteration from the same initial guess)."""
    M, nl, nx, nu, off, n_p = chain_layout(n_mass)
    ode = make_chain_ode(n_mass)
    pt = torch.tensor(p)
    pos0 = np.zeros((M + 1, 3))
    pos0[:, 0] = np.linspace(0.0, x_end[0], M + 2)[1:]
    pos0[M] = x_end
    free = torch.tensor(pos0[:M].reshape(-1))

    def resid(fr):
        # ode output = [vel (3M), u (3), f (3M)]  -> accelerations are the last 3M entries
        x = torch.cat([fr, torch.tensor(x_end), torch.zeros(3 * M, dtype=DT)])
        return ode(x, torch.zeros(3, dtype=DT), pt)[3 * M + 3:]

    for _ in range(100):
        r = resid(free)
        J = torch.autograd.functional.jacobian(resid, free)
        step = torch.linalg.solve(J, -r)
        free = free + step
        if float(step.abs().max()) < 1e-15:
            break
    x_ss = np.concatenate([free.numpy(), x_end, np.zeros(3 * M)])
    assert float(resid(free).abs().max()) < 1e-10
    return x_ss


def make_chain_mass(n_mass: int = 5, N: int = 40, Ts: float = 0.2) -> Problem:
    M, nl, nx, nu, off, n_p = chain_layout(n_mass)
    ode = make_chain_ode(n_mass)
    m0, D0, L0, C0 = 0.033, 1.0, 0.033, 0.1                   # ocp_utils.py:330-333 (random_scale = 0)
    p0 = np.zeros(n_p)
    p0[off["m"][0]: off["m"][1]] = m0
    p0[off["D"][0]: off["D"][1]] = D0
    p0[off["L"][0]: off["L"][1]] = L0
    p0[off["C"][0]: off["C"][1]] = C0
    q_diag = np.ones(nx)
    q_diag[3 * M: 3 * M + 3] = M + 1                          # ocp_utils.py:268-270
    Q = 2.0 * np.diag(q_diag)
    R = 2.0 * 1e-2 * np.eye(nu)                               # ocp_utils.py:274
    p0[off["Q"][0]: off["Q"][1]] = Q.flatten("F")
    p0[off["R"][0]: off["R"][1]] = R.flatten("F")
    x_end = np.array([L0 * (n_mass - 1) * 6, 0.0, 0.0])       # ocp_utils.py:249
    x_ss = chain_steady_state(n_mass, p0, x_end)
    xss_t = torch.tensor(x_ss)
    labels = [f"m_{i}" for i in range(nl)]
    for key in ("D", "L", "C"):
        labels += [f"{key}_{i}_{j}" for i in range(nl) for j in range(3)]
    labels += [f"Q_{i}" for i in range(nx * nx)] + [f"R_{i}" for i in range(nu * nu)]
    labels += [f"w_{i}_{j}" for i in range(M) for j in range(3)]

    def F(x, u, p):
        return rk4(ode, x, u, p, Ts / 2, 2)                    # 2 RK4 steps of Ts/2 (ocp_utils.py:42-56,132)

    def Qm(p):
        return p[off["Q"][0]: off["Q"][1]].reshape(nx, nx).T   # column-major reshape (ocp_utils.py:267)

    def Rm(p):
        return p[off["R"][0]: off["R"][1]].reshape(nu, nu).T

    def stage_cost(k, x, u, p):
        e = x - xss_t
        return 0.5 * (e @ (Qm(p) @ e) + u @ (Rm(p) @ u))       # ocp_utils.py:276

    def terminal_cost(x, p):
        e = x - xss_t
        return 0.5 * (e @ (Qm(p) @ e))                         # ocp_utils.py:277

    # examples/chain_mass.py:17-25
    x0 = np.zeros(nx)
    x0[: 3 * (M + 1): 3] = np.linspace(0.0, L0 * (M + 1) * 6, M + 2)[1:]
    return Problem(
        name=f"chain_mass_{n_mass}", nx=nx, nu=nu, N=N, dT=Ts, p0=p0, p_labels=labels,
        F=F, stage_cost=stage_cost, terminal_cost=terminal_cost, cost_kind="EXTERNAL",
        lbu=-np.ones(nu), ubu=np.ones(nu), gamma=1.0, tol=1e-5, max_iter=50, x0_default=x0,
        extra={"x_ss": x_ss, "off": off, "n_mass": n_mass, "M": M, "tol_mu_factor": 1.0},
    )


def make_problem(name: str, **kw) -> Problem:
    if name == "cartpole":
        return make_cartpole(**kw)
    if name == "linear_system":
        return make_linear_system(**kw)
    if name.startswith("chain_mass"):
        return make_chain_mass(**kw)
    raise KeyError(name)
