"""Problem descriptors: the reference's OCP definitions restated as plain data for the C ABI.

Reference anchors
  cartpole        rlmpc/mpc/cartpole/acados.py:28-108,162-203 + config/cartpole.yaml
  linear system   rlmpc/mpc/linear_system/acados.py:73-131, tests/test_linear_example.py:9-17
The parameter vector ``p`` keeps the reference's order (rlmpc/mpc/nlp.py:969-989): ``model`` block, then
W_0, W, W_e, yref_0, yref, yref_e (column-major) when the cost has them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib

NO_BOUND = _lib.NO_BOUND


@dataclass
class OcpDescription:
    name: str
    model: int
    N: int
    nx: int
    nu: int
    dT: float
    cost_kind: int
    h: float
    rk_steps: int
    p0: np.ndarray
    p_labels: List[str]
    x_labels: List[str]
    u_labels: List[str]
    lbu: np.ndarray
    ubu: np.ndarray
    idxbx: np.ndarray = field(default_factory=lambda: np.zeros(0, int))
    lbx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0, int))
    lbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxsbx: np.ndarray = field(default_factory=lambda: np.zeros(0, int))
    zl: np.ndarray = field(default_factory=lambda: np.zeros(0))
    zu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    consts: np.ndarray = field(default_factory=lambda: np.zeros(0))
    gamma: float = 1.0
    tol: float = 1e-6
    max_iter: int = 500
    x0: Optional[np.ndarray] = None
    n_model_p: int = 0            # length of the ``model`` block of p (what ocp.dims.np is in the reference)
    cost_fields: dict = field(default_factory=dict)   # name -> (offset, shape) of W_0, W, ... inside p

    @property
    def n_p(self) -> int:
        return int(self.p0.shape[0])

    def stage_bounds(self):
        """Box bounds in stage-vector order v = [u; x]; +-NO_BOUND = absent."""
        nw = self.nu + self.nx
        lb, ub = np.full(nw, -NO_BOUND), np.full(nw, NO_BOUND)
        lb[: self.nu], ub[: self.nu] = self.lbu, self.ubu
        soft = np.zeros(nw, np.int32)
        zl, zu = np.zeros(nw), np.zeros(nw)
        for j, ix in enumerate(self.idxbx):
            lb[self.nu + ix], ub[self.nu + ix] = self.lbx[j], self.ubx[j]
        for n, j in enumerate(self.idxsbx):
            ix = self.idxbx[j]
            soft[self.nu + ix] = 1
            zl[self.nu + ix], zu[self.nu + ix] = self.zl[n], self.zu[n]
        lbe, ube = np.full(self.nx, -NO_BOUND), np.full(self.nx, NO_BOUND)
        for j, ix in enumerate(self.idxbx_e):
            lbe[ix], ube[ix] = self.lbx_e[j], self.ubx_e[j]
        return lb, ub, lbe, ube, soft, zl, zu

    def c_spec(self):
        """(ProblemSpec, keep-alive list) for mpcrl_create."""
        lb, ub, lbe, ube, soft, zl, zu = self.stage_bounds()
        lb0, ub0 = lb[: self.nu].copy(), ub[: self.nu].copy()
        consts = np.ascontiguousarray(self.consts, float)
        keep = [lb, ub, lbe, ube, soft, zl, zu, lb0, ub0, consts]
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        spec = _lib.ProblemSpec(
            model=self.model, N=self.N, nx=self.nx, nu=self.nu, np=self.n_p, cost_kind=self.cost_kind, dT=self.dT,
            gamma=self.gamma, h=self.h, rk_steps=self.rk_steps, tol=self.tol, max_iter=self.max_iter,
            lb0=dp(lb0), ub0=dp(ub0), lb=dp(lb), ub=dp(ub), lbe=dp(lbe), ube=dp(ube),
            soft=soft.ctypes.data_as(C.POINTER(C.c_int32)), zl=dp(zl), zu=dp(zu), consts=dp(consts), n_consts=len(consts))
        return spec, keep


def cartpole_ocp(N: int = 20, tf: float = 2.0, M: float = 1.0, m: float = 0.1, l: float = 0.5,
                 W=None, W_e=None, yref=None, yref_e=None, W_0=None, yref_0=None, max_iter: int = 500,
                 tol: float = 1e-6) -> OcpDescription:
    """Cartpole swing-up OCP.  Defaults: config/cartpole.yaml with the horizon of BASELINE.json (N=20 at the
    reference's dt = tf/N = 0.1; the yaml itself has N=30, tf=3.0).  One RK4 step of h = tf/N/4 per stage
    (rlmpc/mpc/cartpole/acados.py:86-92)."""
    W = np.diag([10.0, 0.1, 10.0, 0.1, 0.01]) if W is None else np.asarray(W, float)
    W_e = np.diag([10.0, 0.1, 10.0, 0.1]) if W_e is None else np.asarray(W_e, float)
    yref = np.zeros(5) if yref is None else np.asarray(yref, float)
    yref_e = np.zeros(4) if yref_e is None else np.asarray(yref_e, float)
    W_0 = W if W_0 is None else np.asarray(W_0, float)                 # yaml: W_0 = W, yref_0 = yref
    yref_0 = yref if yref_0 is None else np.asarray(yref_0, float)
    # the cost block of p is what the kernels use (csrc/models_dev.hpp CartpoleDev): set_parameter / cost_set reach the solve
    p0 = np.concatenate([[M, m, l], W_0.flatten("F"), W.flatten("F"), W_e.flatten("F"), yref_0, yref, yref_e])
    labels = ["M", "m", "l"] + [f"W_0_{i}" for i in range(25)] + [f"W_{i}" for i in range(25)] + \
        [f"W_e_{i}" for i in range(16)] + [f"yref_0_{i}" for i in range(5)] + [f"yref_{i}" for i in range(5)] + \
        [f"yref_e_{i}" for i in range(4)]
    xb = np.array([2.4, 10.0, 6.28, 10.0])
    fields = {"W_0": (3, (5, 5)), "W": (28, (5, 5)), "W_e": (53, (4, 4)), "yref_0": (69, (5,)), "yref": (74, (5,)),
              "yref_e": (79, (4,))}
    return OcpDescription(
        name="cartpole", model=_lib.MODEL_CARTPOLE, N=N, nx=4, nu=1, dT=tf / N, cost_kind=_lib.COST_NLS, h=tf / N / 4,
        rk_steps=1, p0=p0, p_labels=labels, x_labels=["x", "x_dot", "theta", "theta_dot"], u_labels=["F"],
        lbu=np.array([-30.0]), ubu=np.array([30.0]), idxbx=np.arange(4), lbx=-xb, ubx=xb, idxbx_e=np.arange(4),
        lbx_e=-xb, ubx_e=xb, tol=tol, max_iter=max_iter, x0=np.array([0.0, 0.0, 3.14, 0.0]), n_model_p=3, cost_fields=fields)


def linear_system_ocp(param: Optional[dict] = None, discount_factor: float = 0.99, N: int = 40) -> OcpDescription:
    """2-state LTI OCP of rlmpc/mpc/linear_system/acados.py:73-131 (``param`` as in tests/test_linear_example.py:9-17)."""
    from scipy.linalg import solve_discrete_are

    if param is None:
        param = {"A": np.array([[1.0, 0.25], [0.0, 1.0]]), "B": np.array([[0.03125], [0.25]]), "Q": np.identity(2),
                 "R": np.identity(1), "b": np.array([[0.0], [0.0]]), "f": np.array([[0.0], [0.0], [0.0]]),
                 "V_0": np.array([1e-3])}
    A, B = np.asarray(param["A"], float), np.asarray(param["B"], float)
    P = solve_discrete_are(A, B, np.asarray(param["Q"], float), np.asarray(param["R"], float))
    p0 = np.concatenate([np.asarray(param[key], float).T.reshape(-1) for key in ["A", "B", "b", "V_0", "f"]])
    labels = ["A_0", "A_1", "A_2", "A_3", "B_0", "B_1", "b_0", "b_1", "V_0", "f_0", "f_1", "f_2"]
    return OcpDescription(
        name="linear_system", model=_lib.MODEL_LINEAR, N=N, nx=2, nu=1, dT=1.0, cost_kind=_lib.COST_EXTERNAL, h=0.0,
        rk_steps=0, p0=p0, p_labels=labels, x_labels=["x_0", "x_1"], u_labels=["u"],
        lbu=np.array([-1.0]), ubu=np.array([1.0]), idxbx=np.arange(2), lbx=np.array([0.0, -1.0]), ubx=np.array([1.0, 1.0]),
        idxsbx=np.array([0]), zl=np.array([1e2]), zu=np.array([1e2]), consts=P.reshape(-1).copy(), gamma=discount_factor,
        tol=1e-6, max_iter=100, x0=np.array([0.5, 0.5]), n_model_p=12)


# ---------------------------------------------------------------------------------------------------
# chain of masses (rlmpc/mpc/chain_mass/ocp_utils.py:59-147,195-376, rlmpc/examples/chain_mass.py:17-25)
# ---------------------------------------------------------------------------------------------------
def chain_param_layout(n_mass: int):
    """Offsets of m, D, L, C, Q, R, w inside p (define_param_struct_symSX, ocp_utils.py:353-371)."""
    M, nl = n_mass - 2, n_mass - 1
    nx, nu = (2 * M + 1) * 3, 3
    off, o = {}, 0
    for key, sz in [("m", nl), ("D", 3 * nl), ("L", 3 * nl), ("C", 3 * nl), ("Q", nx * nx), ("R", nu * nu), ("w", 3 * M)]:
        off[key] = (o, o + sz)
        o += sz
    return M, nl, nx, nu, off, o


def _chain_accel(n_mass, pos, vel, u, p):
    """Accelerations of the free masses (ocp_utils.py:72-125), numpy."""
    M, nl, nx, nu, off, _ = chain_param_layout(n_mass)
    m = p[off["m"][0]: off["m"][1]]
    D = p[off["D"][0]: off["D"][1]].reshape(nl, 3)
    L = p[off["L"][0]: off["L"][1]].reshape(nl, 3)
    C = p[off["C"][0]: off["C"][1]].reshape(nl, 3)
    w = p[off["w"][0]: off["w"][1]].reshape(M, 3)
    dist = pos - np.vstack([np.zeros((1, 3)), pos[:-1]])
    nrm = np.sqrt((dist * dist).sum(axis=1, keepdims=True))
    Fs = D / m[:, None] * (1.0 - L / nrm) * dist
    dv = np.vstack([vel, u[None, :]]) - np.vstack([np.zeros((1, 3)), vel])
    Ft = Fs + C * dv
    return -Ft[:M] + Ft[1:] + np.array([0.0, 0.0, -9.81])[None, :] + w


def chain_steady_state(n_mass: int, p: np.ndarray, x_end: np.ndarray) -> np.ndarray:
    """x_ss with f = 0, u = 0 and the last mass at x_end (compute_parametric_steady_state, ocp_utils.py:150-192;
    the reference mis-uses IPOPT as a root finder, here: Newton with a finite-difference Jacobian)."""
    M = n_mass - 2
    pos = np.zeros((M + 1, 3))
    pos[:, 0] = np.linspace(0.0, x_end[0], M + 2)[1:]
    pos[M] = x_end

    def resid(free):
        q = pos.copy()
        q[:M] = free.reshape(M, 3)
        return _chain_accel(n_mass, q, np.zeros((M, 3)), np.zeros(3), p).reshape(-1)

    free = pos[:M].reshape(-1).copy()
    for _ in range(100):
        r = resid(free)
        J = np.empty((3 * M, 3 * M))
        for j in range(3 * M):
            e = np.zeros(3 * M)
            e[j] = 1e-7
            J[:, j] = (resid(free + e) - resid(free - e)) / 2e-7
        step = np.linalg.solve(J, -r)
        free = free + step
        if np.abs(step).max() < 1e-15:
            break
    assert np.abs(resid(free)).max() < 1e-10, "chain steady state did not converge"
    return np.concatenate([free, x_end, np.zeros(3 * M)])


def chain_mass_ocp(n_mass: int = 5, N: int = 40, Ts: float = 0.2, m: float = 0.033, D: float = 1.0, L: float = 0.033,
                   C: float = 0.1, max_iter: int = 50, tol: float = 1e-5) -> OcpDescription:
    """Chain-of-masses OCP with get_chain_params() defaults (ocp_utils.py:319-341; random_scale = 0)."""
    M, nl, nx, nu, off, n_p = chain_param_layout(n_mass)
    p0 = np.zeros(n_p)
    for key, val in (("m", m), ("D", D), ("L", L), ("C", C)):
        p0[off[key][0]: off[key][1]] = val
    q_diag = np.ones(nx)
    q_diag[3 * M: 3 * M + 3] = M + 1                                  # ocp_utils.py:268-270
    p0[off["Q"][0]: off["Q"][1]] = (2.0 * np.diag(q_diag)).flatten("F")
    p0[off["R"][0]: off["R"][1]] = (2.0 * 1e-2 * np.eye(nu)).flatten("F")   # ocp_utils.py:274
    x_end = np.array([L * (n_mass - 1) * 6, 0.0, 0.0])               # ocp_utils.py:249
    x_ss = chain_steady_state(n_mass, p0, x_end)
    labels = [f"m_{i}" for i in range(nl)]
    for key in ("D", "L", "C"):
        labels += [f"{key}_{i}_{j}" for i in range(nl) for j in range(3)]
    labels += [f"Q_{i}" for i in range(nx * nx)] + [f"R_{i}" for i in range(nu * nu)] + [f"w_{i}_{j}" for i in range(M) for j in range(3)]
    x0 = np.zeros(nx)
    x0[: 3 * (M + 1): 3] = np.linspace(0.0, L * (M + 1) * 6, M + 2)[1:]   # examples/chain_mass.py:17-25
    return OcpDescription(
        name=f"chain_mass_{n_mass}", model=_lib.MODEL_CHAIN, N=N, nx=nx, nu=nu, dT=Ts, cost_kind=_lib.COST_EXTERNAL, h=Ts / 2,
        rk_steps=2, p0=p0, p_labels=labels, x_labels=[f"x_{i}" for i in range(nx)], u_labels=[f"u_{i}" for i in range(nu)],
        lbu=-np.ones(nu), ubu=np.ones(nu), consts=x_ss, gamma=1.0, tol=tol, max_iter=max_iter, x0=x0, n_model_p=n_p)
