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


class _DM(np.ndarray):
    """ndarray with the two CasADi ``DM`` methods the reference's call sites use (``.full()``, ``float()``)."""

    def __new__(cls, a):
        return np.asarray(a, float).view(cls)

    def full(self) -> np.ndarray:
        return np.array(self, dtype=float).reshape(self.shape if self.ndim == 2 else (-1, 1))


class _PStruct:
    """``nlp.p.sym(value)`` / ``nlp.p.val`` look-alike: the full parameter vector viewed as the struct of nlp.py:969-989
    (``model`` block, then W_0, W, W_e, yref_0, yref, yref_e — each column-major)."""

    def __init__(self, ocp: OcpDescription, vec):
        self._ocp = ocp
        self._v = np.asarray(vec, float).reshape(-1).copy()
        if self._v.shape[0] != ocp.n_p:
            raise ValueError(f"parameter vector has {self._v.shape[0]} entries, expected {ocp.n_p}")

    def _span(self, key):
        if key == "model":
            return 0, (self._ocp.n_model_p,)
        return self._ocp.cost_fields[key]

    def keys(self):
        return ["model"] + list(self._ocp.cost_fields.keys())

    def __getitem__(self, key) -> _DM:
        off, shape = self._span(key)
        return _DM(self._v[off: off + int(np.prod(shape))].reshape(shape, order="F"))

    def __setitem__(self, key, value):
        off, shape = self._span(key)
        self._v[off: off + int(np.prod(shape))] = np.asarray(value, float).flatten("F")

    @property
    def cat(self) -> _DM:
        return _DM(self._v.reshape(-1, 1))


class _ParamView:
    """``mpc.nlp.p``: ``.val.cat.full()`` -> (n_p, 1) (mpc.py:160), ``.val[key]``, ``.sym(value)`` (mpc.py:214,234)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def val(self) -> _PStruct:
        return _PStruct(self._o.ocp, self._o._p)

    def sym(self, value) -> _PStruct:
        return _PStruct(self._o.ocp, value)


class _VarsVal:
    """``mpc.nlp.vars.val`` look-alike (nlp.py:903-964,1137-1166): item access to the primal iterate, the per-stage bound
    parameters, ``dT`` and ``gamma`` — reads come from the engine, writes are routed to it."""

    _BOUND = ("lbu", "ubu", "lbx", "ubx")

    def __init__(self, owner):
        self._o = owner

    @staticmethod
    def _split(key):
        if isinstance(key, tuple):
            return key[0], int(key[1])
        name, _, stage = key.rpartition("_")
        if name in _VarsVal._BOUND and stage.isdigit():
            return name, int(stage)
        return key, None

    def __getitem__(self, key):
        o = self._o
        f, k = self._split(key)
        if f in ("x", "u"):
            return _DM(o.get(k, f))
        if f == "dT":
            return o.ocp.dT
        if f == "gamma":
            return o.discount_factor
        if f == "p":
            return _DM(o.get_p())
        if f in self._BOUND:
            return _DM(o._bound_value(k, f))
        raise KeyError(key)

    def __setitem__(self, key, value):
        f, k = self._split(key)
        if f == "gamma":
            self._o.set_discount_factor(float(value))
        else:
            self._o.nlp.set(k, f, value)


class _NlpShim:
    """The ``mpc.nlp`` members the reference's MPC class and scripts touch."""

    def __init__(self, owner):
        self._o = owner
        self.p = _ParamView(owner)
        self.vars = SimpleNamespace(val=_VarsVal(owner))

    @property
    def dL_dp(self):
        return SimpleNamespace(val=SimpleNamespace(full=lambda: self._o.get_dL_dp()))

    @property
    def dpi_dp(self):
        return SimpleNamespace(val=self._o.get_dpi_dp())

    @property
    def cost(self):
        return SimpleNamespace(val=self._o.get_V())

    @property
    def L(self):
        return SimpleNamespace(val=self._o.get_L())

    def get_parameter(self, field_) -> _DM:          # nlp.py:317-318
        return _PStruct(self._o.ocp, self._o._p)[field_]

    def set_parameter(self, field_, value_):          # nlp.py:314-315
        ps = _PStruct(self._o.ocp, self._o._p)
        ps[field_] = value_
        self._o._set_p_internal(ps._v)

    def set(self, stage_, field_, value_):            # nlp.py:285-312
        value_ = np.asarray(value_, float).reshape(-1)
        if len(value_) == 0:
            return
        o = self._o
        if field_ in ("x", "u", "pi"):
            o.set(stage_, field_, value_)
        elif field_ == "p":
            o.set(stage_, "p", value_)
        elif field_ == "dT":
            if abs(float(value_[0]) - o.ocp.dT) > 1e-15:
                raise NotImplementedError("dT is fixed by the OcpDescription (tf / N)")
        elif field_ in _VarsVal._BOUND:
            o._set_bound(stage_, field_, value_)
        else:
            raise Exception(f"Field {field_} not supported.")
        return 0

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
        """Solve with the x0 / stage-0 input bounds last written through ``constraints_set`` / ``set`` (lbu_0 = ubu_0 pins u_0)."""
        return self._o._solve(self._o._x0, self._o._u0_pin(), sens=False)

    def reset(self):
        self._o._batch.reset()

    def cost_set(self, stage, field, value, api="new"):
        """cost_set(stage, "W" | "yref", value) (mpc.py:236-252).  The reference writes stage 0 from W_0 / yref_0, EVERY stage
        1..N-1 from W / yref and never touches the terminal stage; the engine keeps one W / yref for all interior stages, so a
        call for any of them updates that one."""
        key = {"W": "W_0" if stage == 0 else ("W_e" if stage == self._o.ocp.N else "W"),
               "yref": "yref_0" if stage == 0 else ("yref_e" if stage == self._o.ocp.N else "yref")}[field]
        if key not in self._o.ocp.cost_fields:
            raise Exception(f"this OCP has no cost field {key}")
        self._o.nlp.set_parameter(key, value)

    def constraints_set(self, stage, field, value):
        """constraints_set(stage, "lbu" | "ubu" | "lbx" | "ubx", value) (mpc.py:72-73,87-88)."""
        self._o._set_bound(stage, field, np.asarray(value, float).reshape(-1))


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
        # box bounds as the solver currently has them (stage-vector order v = [u; x]); constraints_set / nlp.set edit them
        lb, ub, lbe, ube, _, _, _ = ocp.stage_bounds()
        self._lb0, self._ub0 = lb[: ocp.nu].copy(), ub[: ocp.nu].copy()
        self._lb, self._ub, self._lbe, self._ube = lb, ub, lbe, ube
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

    def _u0_pin(self):
        """lbu_0 == ubu_0 written through constraints_set is the reference's way of pinning u_0 (Q(s, a), mpc.py:71-76)."""
        return self._lb0.copy() if np.array_equal(self._lb0, self._ub0) else None

    def _bound_value(self, stage, field):
        ocp, nu = self.ocp, self.ocp.nu
        lo = field.startswith("l")
        if field in ("lbu", "ubu"):
            return (self._lb0 if lo else self._ub0).copy() if stage == 0 else (self._lb if lo else self._ub)[:nu].copy()
        if stage == 0:
            return None if self._x0 is None else self._x0.copy()          # lbx_0 = ubx_0 = x0 (mpc.py:38-39)
        if stage == ocp.N:
            return (self._lbe if lo else self._ube)[ocp.idxbx_e].copy()
        return (self._lb if lo else self._ub)[nu + ocp.idxbx].copy()

    def _set_bound(self, stage, field, value):
        """One bound vector of one stage.  Stage 0: lbu/ubu are the stage-0 input bounds (equal = u_0 pinned), lbx/ubx are x0.
        Stages 1..N-1 share one set of bounds (as ocp.constraints.lbx/ubx/lbu/ubu do in the reference), stage N has its own."""
        from . import _lib
        ocp, nu = self.ocp, self.ocp.nu
        value = np.asarray(value, float).reshape(-1)
        lo = field.startswith("l")
        self._sens_fresh = False               # whichever bound changes, the cached sensitivities belong to the old problem
        if field in ("lbx", "ubx") and stage == 0:
            self._x0 = value.copy()
            return
        if field in ("lbu", "ubu"):
            if stage == 0:
                (self._lb0 if lo else self._ub0)[:] = value
                if np.all(self._lb0 < self._ub0):      # a pin (lb == ub) is passed per solve as u0_fixed instead
                    self._batch.set_bounds(_lib.BOUNDS_U0, self._lb0, self._ub0)
                return
            if stage >= ocp.N:
                raise Exception("no input at the terminal stage")
            (self._lb if lo else self._ub)[:nu] = value
        elif field in ("lbx", "ubx"):
            if stage == ocp.N:
                (self._lbe if lo else self._ube)[ocp.idxbx_e] = value
                if np.all(self._lbe <= self._ube):
                    self._batch.set_bounds(_lib.BOUNDS_TERMINAL, self._lbe, self._ube)
                return
            (self._lb if lo else self._ub)[nu + ocp.idxbx] = value
        else:
            raise Exception(f"Field {field} not supported.")
        if np.all(self._lb <= self._ub):
            self._batch.set_bounds(_lib.BOUNDS_STAGE, self._lb, self._ub)

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
        self._x0 = np.asarray(x0, float).reshape(-1).copy()
        self._batch.reset(self._x0.reshape(1, -1))
        self._batch.set_discount_factor(self.discount_factor)

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
        if field not in ("x", "u", "pi"):
            raise Exception(f"Field {field} not supported.")
        # Before the first solve and after reset() the handle holds the cold iterate (x_k = x0 or 0, u = 0, multipliers 0,
        # slacks 1); the edited stage goes on top of it and the bound multipliers stay cold (bnd = None), so that the
        # reference's initialisation pattern `for stage: ocp_solver.set(stage, "x", x0)` (mpc.py:208-210) is a cold start
        # from that guess — never a warm start from stale multipliers.
        duals = self._batch.duals_valid
        x, u, pi, bnd, _ = self._batch.get_iterate()
        tgt = {"x": x, "u": u, "pi": pi}[field]
        tgt[0, stage] = torch.as_tensor(np.asarray(value, float).reshape(-1), device=tgt.device)
        self._batch.set_iterate(x, u, pi, bnd if duals else None)

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

    def get_L(self) -> float:                         # mpc.py:325-332 (float(nlp.L.val), nlp.py:1180,1390)
        return float(self._batch.get_lagrangian()[0].item())

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
