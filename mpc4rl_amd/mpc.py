"""MPC — the reference's single-instance policy / value-function surface on top of the batched engine.

Same method names, argument meaning, return shapes and error behaviour as ``rlmpc.mpc.common.mpc.MPC``
(rlmpc/mpc/common/mpc.py:8-414) so that ``rlmpc.examples.linear_system_mpc_qlearning`` style drivers run
unchanged; internally it is an ``MPCBatch`` of size 1.  Differences, on purpose:

* ``update`` / ``get_action`` leave the sensitivities stale in the reference until ``update_nlp`` is called
  (quirk q3, mpc.py:177-202); here ``get_dV_dp`` / ``get_dpi_dp`` refresh them on demand.
* ``update_nlp`` prints nothing (the reference prints 12 lines per call, nlp.py:1541-1553).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .batch import MPCBatch
from .problems import OcpDescription, cartpole_ocp, chain_mass_ocp, linear_system_ocp


class _ParamView:
    """``mpc.nlp.p`` look-alike: ``.val.cat.full()`` -> (n_p, 1) array (mpc.py:160)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def val(self):
        o = self._o
        return SimpleNamespace(cat=SimpleNamespace(full=lambda: o._p.copy().reshape(-1, 1)))


class _NlpShim:
    """The handful of ``mpc.nlp`` members the reference's scripts touch."""

    def __init__(self, owner):
        self._o = owner
        self.p = _ParamView(owner)

    @property
    def dL_dp(self):
        return SimpleNamespace(val=SimpleNamespace(full=lambda: self._o.get_dL_dp()))

    @property
    def dpi_dp(self):
        return SimpleNamespace(val=self._o.get_dpi_dp())

    @property
    def cost(self):
        return SimpleNamespace(val=self._o.get_V())

    def get_parameter(self, field_):
        off, shape = self._o.ocp.cost_fields[field_]
        n = int(np.prod(shape))
        return self._o._p[off: off + n].reshape(shape, order="F")

    def set_parameter(self, field_, value_):
        off, shape = self._o.ocp.cost_fields[field_]
        p = self._o._p.copy()
        p[off: off + int(np.prod(shape))] = np.asarray(value_, float).flatten("F")
        self._o._set_p_internal(p)

    def assert_kkt_residual(self, tol: float = 1e-6) -> bool:
        """nlp.py:1295-1299: all four KKT residual norms below tol."""
        res = self._o._batch.get_iterate()[4][0].cpu().numpy()
        assert np.all(res < tol), f"KKT residual {res} >= {tol}"
        return True


class _SolverShim:
    """``mpc.ocp_solver`` look-alike (AcadosOcpSolver call sites of SURVEY.md §2 row 3)."""

    def __init__(self, owner):
        self._o = owner
        ocp = owner.ocp
        self.acados_ocp = SimpleNamespace(
            dims=SimpleNamespace(N=ocp.N, nx=ocp.nx, nu=ocp.nu, np=ocp.n_model_p),
            constraints=SimpleNamespace(lbx=ocp.lbx, ubx=ocp.ubx, lbu=ocp.lbu, ubu=ocp.ubu, idxbx=ocp.idxbx,
                                        idxbu=np.arange(ocp.nu), lbx_0=None if ocp.x0 is None else ocp.x0.copy(),
                                        ubx_0=None if ocp.x0 is None else ocp.x0.copy()),
            parameter_values=ocp.p0[: ocp.n_model_p].copy(),
            cost=SimpleNamespace(**{k: ocp.p0[o: o + int(np.prod(s))].reshape(s, order="F") for k, (o, s) in ocp.cost_fields.items()}),
            solver_options=SimpleNamespace(tf=ocp.dT * ocp.N),
            model=SimpleNamespace(name=ocp.name),
        )

    @property
    def status(self):
        return self._o.status

    def get(self, stage: int, field: str):
        return self._o.get(stage, field)

    def set(self, stage: int, field: str, value):
        return self._o.set(stage, field, value)

    def get_cost(self) -> float:
        return self._o.get_V()

    def get_residuals(self):
        return self._o._batch.get_iterate()[4][0].cpu().numpy()

    def solve(self) -> int:
        return self._o._solve(self._o._x0, None, sens=False)

    def reset(self):
        self._o._batch.reset()

    def cost_set(self, stage, field, value, api="new"):
        key = {"W": "W_0" if stage == 0 else ("W_e" if stage == self._o.ocp.N else "W"),
               "yref": "yref_0" if stage == 0 else ("yref_e" if stage == self._o.ocp.N else "yref")}[field]
        self._o.nlp.set_parameter(key, value)

    def constraints_set(self, stage, field, value):
        raise NotImplementedError("bounds are part of the OcpDescription; use q_update(x0, u0) to pin u_0")


class MPC:
    """MPC-as-policy / value function around one OCP instance (rlmpc/mpc/common/mpc.py:8)."""

    def __init__(self, ocp: OcpDescription, gamma: Optional[float] = None, device=None):
        self.ocp = ocp
        self._batch = MPCBatch(ocp, 1, device)
        self.discount_factor = ocp.gamma if gamma is None else gamma
        self._batch.set_discount_factor(self.discount_factor)
        self.nlp_timing = {}
        self.status = 0
        self._p = ocp.p0.copy()
        self._x0 = None if ocp.x0 is None else ocp.x0.copy()
        self._u0 = None
        self._last = None
        self._sens_fresh = False
        self.ocp_solver = _SolverShim(self)
        self.nlp = _NlpShim(self)

    # ---------------------------------------------------------------- internals
    def _solve(self, x0, u0, sens: bool) -> int:
        x0 = np.asarray(x0, float).reshape(1, self.ocp.nx)
        u0t = None if u0 is None else np.asarray(u0, float).reshape(1, self.ocp.nu)
        r = self._batch.solve(x0, u0t, sens_v=sens, sens_pi=sens)
        self._x0, self._u0 = x0[0].copy(), None if u0 is None else u0t[0].copy()
        self._last = r
        self._sens_fresh = sens
        self.status = int(r.status[0].item())
        return self.status

    def _set_p_internal(self, p):
        self._p = np.asarray(p, float).reshape(-1).copy()
        self._batch.set_theta(torch.as_tensor(self._p))
        self._sens_fresh = False

    # ---------------------------------------------------------------- reference surface
    def get_parameters(self) -> np.ndarray:          # mpc.py:24
        return self.get_p()

    def get_action(self, x0: np.ndarray) -> np.ndarray:   # mpc.py:27-50 (status stored silently)
        self._solve(x0, None, sens=False)
        return self._last.u0[0].cpu().numpy()

    def q_update(self, x0: np.ndarray, u0: np.ndarray) -> int:   # mpc.py:52-96
        status = self._solve(x0, u0, sens=True)
        if status != 0:
            raise RuntimeError(f"Solver failed q_update with status {status}. Exiting.")
        return status

    def update_nlp(self) -> None:                     # mpc.py:98-102
        if self._x0 is None:
            raise RuntimeError("update_nlp called before any solve")
        self._solve(self._x0, self._u0, sens=True)

    def get_dV_dp(self) -> np.ndarray:                # mpc.py:104-113
        return self.get_dL_dp()

    def get_Q(self) -> float:                         # mpc.py:115-124
        return float(self._last.V[0].item())

    def get_dQ_dp(self) -> np.ndarray:                # mpc.py:126-135
        return self.get_dL_dp()

    def set_p(self, p: np.ndarray, finite_differences: bool = False) -> None:   # mpc.py:137-154
        self._set_p_internal(p)

    def get_p(self) -> np.ndarray:                    # mpc.py:156-160
        return self._p.copy()

    def get_parameter_values(self) -> np.ndarray:     # mpc.py:162-166
        return self.get_p()

    def get_parameter_labels(self) -> list:           # mpc.py:168-169 (labels of ocp.model.p)
        return list(self.ocp.p_labels[: self.ocp.n_model_p])

    def get_state_labels(self) -> list:               # mpc.py:171-172
        return list(self.ocp.x_labels)

    def get_input_labels(self) -> list:               # mpc.py:174-175
        return list(self.ocp.u_labels)

    def update(self, x0: np.ndarray) -> int:          # mpc.py:177-202
        status = self._solve(x0, None, sens=False)
        if status != 0:
            raise RuntimeError(f"Solver failed update with status {status}. Exiting.")
        return status

    def reset(self, x0: np.ndarray):                  # mpc.py:204-210
        self._batch.reset()
        self._batch.set_discount_factor(self.discount_factor)
        self._x0 = np.asarray(x0, float).reshape(-1).copy()

    def set(self, stage, field, value, finite_differences: bool = False):   # mpc.py:212-231
        if field == "p":
            v = np.asarray(value, float).reshape(-1)
            if v.shape[0] == self.ocp.n_p:
                self._set_p_internal(v)
            else:                                      # only the model block, as ocp_solver.set(stage, "p", ..) takes
                p = self._p.copy()
                p[: self.ocp.n_model_p] = v
                self._set_p_internal(p)
            return
        x, u, pi, bnd, _ = self._batch.get_iterate()
        tgt = {"x": x, "u": u, "pi": pi}.get(field)
        if tgt is None:
            raise Exception(f"Field {field} not supported.")
        tgt[0, stage] = torch.as_tensor(np.asarray(value, float).reshape(-1), device=tgt.device)
        self._batch.set_iterate(x, u, pi, bnd)

    def set_parameter(self, value_, api="new"):       # mpc.py:233-257
        self._set_p_internal(value_)

    def set_discount_factor(self, discount_factor_: float) -> None:   # mpc.py:259-285
        self.discount_factor = discount_factor_
        self._batch.set_discount_factor(discount_factor_)
        self._sens_fresh = False

    def get(self, stage, field):                      # mpc.py:287-288
        x, u, pi, bnd, _ = self._batch.get_iterate()
        nu, nx, N = self.ocp.nu, self.ocp.nx, self.ocp.N
        if field == "x":
            return x[0, stage].cpu().numpy()
        if field == "u":
            return u[0, stage].cpu().numpy()
        if field == "pi":
            return pi[0, stage].cpu().numpy()
        b = bnd[0, :, stage].cpu().numpy()             # [10, nu+nx]
        if stage == 0:
            ub_idx, xb_idx, sb = np.arange(nu), np.zeros(0, int), np.zeros(0, int)
        elif stage < N:
            ub_idx, xb_idx, sb = np.arange(nu), nu + self.ocp.idxbx, nu + self.ocp.idxbx[self.ocp.idxsbx]
        else:
            ub_idx, xb_idx, sb = np.zeros(0, int), nu + self.ocp.idxbx_e, np.zeros(0, int)
        if field in ("lam", "t"):                     # acados order: lbu, lbx, ubu, ubx, lsbx, usbx (common/utils.py:4-25)
            lo, up, slo, sup = (0, 1, 6, 7) if field == "lam" else (2, 3, 8, 9)
            return np.concatenate([b[lo, ub_idx], b[lo, xb_idx], b[up, ub_idx], b[up, xb_idx], b[slo, sb], b[sup, sb]])
        if field == "sl":
            return b[4, sb]
        if field == "su":
            return b[5, sb]
        raise Exception(f"Field {field} not supported.")

    def scale_action(self, action: np.ndarray) -> np.ndarray:     # mpc.py:290-301
        low, high = self.ocp.lbu, self.ocp.ubu
        return 2.0 * ((action - low) / (high - low)) - 1.0

    def unscale_action(self, action: np.ndarray) -> np.ndarray:   # mpc.py:303-314
        low, high = self.ocp.lbu, self.ocp.ubu
        return 0.5 * (high - low) * (action + 1.0) + low

    def get_dL_dp(self) -> np.ndarray:                # mpc.py:316-323 -> (1, n_p)
        if not self._sens_fresh:
            self.update_nlp()
        return self._last.dV_dp.cpu().numpy().reshape(1, -1)

    def get_V(self) -> float:                         # mpc.py:334-343
        return float(self._last.V[0].item())

    def get_pi(self) -> np.ndarray:                   # mpc.py:345-351
        return self._last.u0[0].cpu().numpy()

    def compute_dpi_dp_finite_differences(self, p: np.ndarray, idx: int = None, delta: float = 1e-4) -> np.ndarray:
        """Forward differences, delta = 1e-4 (mpc.py:353-400)."""
        pi0 = self.get_pi().copy()
        p0 = self.get_p()
        x0 = self._x0.copy()
        dpi_dp = np.zeros((self.ocp.nu, p0.shape[0]))
        for i in (range(p0.shape[0]) if idx is None else [idx]):
            pplus = p0.copy()
            pplus[i] += delta
            self.set_p(pplus)
            self.update(x0)
            dpi_dp[:, i] = (self.get_pi() - pi0) / delta
        self.set_p(p0)
        self.update(x0)
        return dpi_dp

    def get_dpi_dp(self, finite_differences: bool = False, idx: int = 0) -> np.ndarray:   # mpc.py:402-414 -> (nu, n_p)
        if finite_differences:
            return self.compute_dpi_dp_finite_differences(self.get_p(), idx=idx)
        if not self._sens_fresh:
            self.update_nlp()
        return self._last.dpi_dp[0].cpu().numpy()


class CartpoleMPC(MPC):
    """rlmpc/mpc/cartpole/acados.py:162-249: get_action returns the action scaled to [-1, 1]."""

    def __init__(self, ocp: Optional[OcpDescription] = None, **kw):
        super().__init__(cartpole_ocp() if ocp is None else ocp, **kw)

    def get_action(self, x0: np.ndarray) -> np.ndarray:   # cartpole/acados.py:239-249
        return self.scale_action(super().get_action(x0))


class LinearSystemMPC(MPC):
    """rlmpc/mpc/linear_system/acados.py:12-24 (``AcadosMPC(param, discount_factor=0.99)``)."""

    def __init__(self, param: Optional[dict] = None, discount_factor: float = 0.99, **kw):
        super().__init__(linear_system_ocp(param, discount_factor), gamma=discount_factor, **kw)


class ChainMassMPC(MPC):
    """rlmpc/mpc/chain_mass/acados.py:18-29 (``AcadosMPC(param, discount_factor)``; param = get_chain_params())."""

    def __init__(self, param: Optional[dict] = None, discount_factor: float = 1.0, **kw):
        param = {} if param is None else param
        ocp = chain_mass_ocp(n_mass=param.get("n_mass", 5), N=param.get("N", 40), Ts=param.get("Ts", 0.2), m=param.get("m", 0.033),
                             D=param.get("D", 1.0), L=param.get("L", 0.033), C=param.get("C", 0.1),
                             max_iter=param.get("nlp_iter", 50), tol=param.get("nlp_tol", 1e-5))
        super().__init__(ocp, gamma=discount_factor, **kw)
