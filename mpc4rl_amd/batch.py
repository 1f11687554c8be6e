"""MPCBatch — the batched extension of the reference's MPC surface: B OCP instances per call.

``rlmpc.td3.policies.Actor.forward`` loops ``mpc.get_action`` over the observation batch in Python
(rlmpc/td3/policies.py:186-197) and the Q-learning example loops ``q_update``/``update`` over replay samples
(rlmpc/examples/linear_system_mpc_qlearning.py:178-190).  Here each of those loops is ONE call that issues one
kernel launch through the C ABI (include/mpcrl.h); inputs and outputs are torch tensors on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from .problems import OcpDescription


@dataclass
class SolveResult:
    u0: torch.Tensor                 # [B, nu]   u_0*
    V: torch.Tensor                  # [B]       optimal cost (V, or Q when u0 was fixed)
    status: torch.Tensor             # [B] int32 0 success, 1 NaN, 2 max-iter, 4 QP failure
    iters: torch.Tensor              # [B, 2] int32  SQP iterations, interior-point iterations
    dV_dp: Optional[torch.Tensor]    # [B, n_p]
    dpi_dp: Optional[torch.Tensor]   # [B, nu, n_p]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class MPCBatch:
    """B independent instances of one OCP on one GPU.  The handle keeps the warm-start iterate of every
    instance (like the reference's solver object keeps its single iterate)."""

    def __init__(self, ocp: OcpDescription, batch: int, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("mpc4rl_amd.MPCBatch needs a HIP device; there is no CPU fallback.")
        self.ocp = ocp
        self.B = int(batch)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:      # "cuda" = the current device, not GPU 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.nx, self.nu, self.N, self.n_p = ocp.nx, ocp.nu, ocp.N, ocp.n_p
        spec, keep = ocp.c_spec()
        h = C.c_void_p()
        rc = self.lib.mpcrl_create(C.byref(spec), self.B, self.device.index, C.byref(h))
        del keep
        if rc != 0:
            raise RuntimeError(f"mpcrl_create failed with {rc}")
        self._h = h
        self._theta = None
        self.has_iterate = False       # a stored iterate exists (after a solve / set_iterate; cleared by reset)
        self.duals_valid = False       # ... and its bound multipliers / slacks come from a solve (not the cold placeholders)
        self.set_theta(torch.as_tensor(ocp.p0, dtype=torch.float64))
        self.gamma = ocp.gamma

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.lib.mpcrl_destroy(h)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, shape) -> torch.Tensor:
        t = torch.as_tensor(t, dtype=torch.float64, device=self.device).reshape(shape).contiguous()
        return t

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed with {rc}")

    # ------------------------------------------------------------------ parameters / options
    def set_theta(self, theta) -> None:
        """theta: [n_p] (shared) or [B, n_p] (per instance).  Mirrors MPC.set_p / set_parameter (mpc.py:137-154,233-257)."""
        t = torch.as_tensor(theta, dtype=torch.float64, device=self.device).contiguous()
        per = int(t.dim() == 2)
        if t.shape[-1] != self.n_p or (per and t.shape[0] != self.B):
            raise ValueError(f"theta must be [{self.n_p}] or [{self.B}, {self.n_p}]")
        self._theta = t
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_set_theta(self._h, _ptr(t), self.n_p, per, self._stream()), "mpcrl_set_theta")

    def get_theta(self) -> torch.Tensor:
        return self._theta

    def set_discount_factor(self, gamma: float) -> None:
        """MPC.set_discount_factor (mpc.py:259-285)."""
        self.gamma = float(gamma)
        self._check(self.lib.mpcrl_set_gamma(self._h, float(gamma)), "mpcrl_set_gamma")

    def set_options(self, tol: float = -1.0, max_iter: int = -1) -> None:
        self._check(self.lib.mpcrl_set_options(self._h, float(tol), int(max_iter)), "mpcrl_set_options")

    def set_exit_rule(self, window: int = 10, factor: float = 0.1) -> None:
        """Opt-in divergence exit (include/mpcrl.h mpcrl_set_exit_rule): every ``window`` SQP iterations the best NLP residual of an
        instance must have dropped below ``factor`` x its value at the previous check, else it ends with status 2.  window = 0: off
        (the reference's behaviour: full-step SQP to max_iter)."""
        self._check(self.lib.mpcrl_set_exit_rule(self._h, int(window), float(factor)), "mpcrl_set_exit_rule")

    def set_launch_mode(self, mode: int = 0) -> None:
        """0 = the library times the time-sliced and the plain launch of the solve kernel against each other on this handle's own
        solves and uses the faster (default), 1 = time-sliced whenever legal, -1 = never (include/mpcrl.h mpcrl_set_launch_mode)."""
        self._check(self.lib.mpcrl_set_launch_mode(self._h, int(mode)), "mpcrl_set_launch_mode")

    def launch_times(self, warm: bool = False):
        """(time-sliced ms, plain ms, preferred shape) of the launch tuner's last probes on cold (default) or warm calls; -1 = not
        measured yet."""
        a, b = C.c_double(-1.0), C.c_double(-1.0)
        rc = self.lib.mpcrl_get_launch_times(self._h, int(warm), C.byref(a), C.byref(b))
        self._check(min(rc, 0), "mpcrl_get_launch_times")
        return a.value, b.value, ("plain" if rc == 1 else "time-sliced")

    def set_bounds(self, which: int, lb, ub) -> None:
        """ocp_solver.constraints_set (mpc.py:72-73,87-88): which = _lib.BOUNDS_U0 (stage-0 controls, nu values), BOUNDS_STAGE
        (stages 1..N-1, v = [u; x], nu + nx values), BOUNDS_TERMINAL (stage N, nx values); |bound| >= 1e29 = absent."""
        lb = np.ascontiguousarray(np.asarray(lb, float).reshape(-1))
        ub = np.ascontiguousarray(np.asarray(ub, float).reshape(-1))
        n = {_lib.BOUNDS_U0: self.nu, _lib.BOUNDS_STAGE: self.nu + self.nx, _lib.BOUNDS_TERMINAL: self.nx}[which]
        if lb.shape[0] != n or ub.shape[0] != n:
            raise ValueError(f"bounds of kind {which} have {n} entries")
        dp = C.POINTER(C.c_double)
        self._check(self.lib.mpcrl_set_bounds(self._h, int(which), lb.ctypes.data_as(dp), ub.ctypes.data_as(dp)), "mpcrl_set_bounds")

    def reset(self, x0=None) -> None:
        """MPC.reset (mpc.py:204-210): the next solve starts from x_k = x0, u = 0, multipliers 0.  With x0 ([B, nx]) the stored
        iterate is set to that cold iterate as well (what get_iterate returns until the next solve)."""
        x0t = None if x0 is None else self._dev(x0, (self.B, self.nx))
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_reset(self._h, _ptr(x0t), self._stream()), "mpcrl_reset")
        self.has_iterate = self.duals_valid = False

    # ------------------------------------------------------------------ the hot path
    def _pack_order(self, x0: torch.Tensor) -> None:
        """Scheduling hint: instances that share a wavefront run in lock-step for the maximum of their iteration counts, so
        neighbours should be similar problems.  Order the batch along the coordinate of x0 with the largest spread."""
        if self.B < 128:
            return
        if 64 // (self.N + 1) < 2:   # one instance per wavefront (the library also skips the chain at any horizon): no lock step to pack for
            self._check(self.lib.mpcrl_set_order(self._h, None, self._stream()), "mpcrl_set_order")
            return
        if self.B <= 8192:   # one small kernel inside the library (csrc/order_kernel.hpp)
            self._check(self.lib.mpcrl_auto_order(self._h, _ptr(x0), self._stream()), "mpcrl_auto_order")
            return
        dim = torch.argmax(x0.max(0).values - x0.min(0).values).reshape(1)     # stays on the device: no host sync
        perm = torch.argsort(torch.index_select(x0, 1, dim).reshape(-1)).to(torch.int32)
        self._check(self.lib.mpcrl_set_order(self._h, _ptr(perm), self._stream()), "mpcrl_set_order")

    def solve(self, x0, u0=None, sens_v: bool = False, sens_pi: bool = False, rti: bool = False, cold: bool = False,
              reorder: bool = True, cold_mask: Optional[torch.Tensor] = None, store_bounds: bool = True,
              exact_qp: bool = False) -> SolveResult:
        """cold_mask [B] (bool / int): these instances ignore their stored iterate (per-environment ``mpc.reset`` at an episode end).
        store_bounds = False (MPCRL_NO_BND_STORE): the bound multipliers / slacks of the solution are not written back to the stored
        iterate — for callers that start every solve cold; the next solve then starts its interior point from the default point.
        exact_qp (MPCRL_EXACT_QP, test-only, cartpole): every QP solved to the tight tolerance, the reference's acados / HPIPM setting."""
        x0 = self._dev(x0, (self.B, self.nx))
        if cold_mask is not None:
            cm = cold_mask.to(device=self.device, dtype=torch.int32).reshape(self.B).contiguous()
            with torch.cuda.device(self.device):
                self._check(self.lib.mpcrl_set_cold_mask(self._h, _ptr(cm), self._stream()), "mpcrl_set_cold_mask")
        u0f = None if u0 is None else self._dev(u0, (self.B, self.nu))
        flags = (_lib.SENS_V if sens_v else 0) | (_lib.SENS_PI if sens_pi else 0) | (_lib.RTI if rti else 0) | \
            (_lib.COLD if cold else 0) | (0 if store_bounds else _lib.NO_BND_STORE) | (_lib.EXACT_QP if exact_qp else 0)
        if reorder:
            if self.lib.mpcrl_query_time_sliced(self._h, flags, self._stream()) == 1:
                # the time-sliced launch deals the batch out in quarters: no packing order needed (and none left over from before)
                self._check(self.lib.mpcrl_set_order(self._h, None, self._stream()), "mpcrl_set_order")
            else:
                with torch.cuda.device(self.device):
                    self._pack_order(x0)
        kw = dict(dtype=torch.float64, device=self.device)
        u0_out = torch.empty((self.B, self.nu), **kw)
        V = torch.empty((self.B,), **kw)
        dV = torch.empty((self.B, self.n_p), **kw) if sens_v else None
        dpi = torch.empty((self.B, self.nu, self.n_p), **kw) if sens_pi else None
        status = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        iters = torch.empty((self.B, 2), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.mpcrl_solve(self._h, _ptr(x0), _ptr(u0f), flags, _ptr(u0_out), _ptr(V), _ptr(dV), _ptr(dpi),
                                      _ptr(status), _ptr(iters), self._stream())
        self._check(rc, "mpcrl_solve")
        self.has_iterate, self.duals_valid = True, bool(store_bounds)
        return SolveResult(u0_out, V, status, iters, dV, dpi)

    def get_action(self, x0) -> torch.Tensor:
        """Batched MPC.get_action (mpc.py:27-50)."""
        return self.solve(x0).u0

    # ------------------------------------------------------------------ iterate access
    def get_iterate(self):
        """x [B,N+1,nx], u [B,N,nu], pi [B,N,nx], bnd [B,10,N+1,nu+nx], res [B,4] (layouts of include/mpcrl.h)."""
        kw = dict(dtype=torch.float64, device=self.device)
        nw = self.nx + self.nu
        x = torch.empty((self.B, self.N + 1, self.nx), **kw)
        u = torch.empty((self.B, self.N, self.nu), **kw)
        pi = torch.empty((self.B, self.N, self.nx), **kw)
        bnd = torch.empty((self.B, 10, self.N + 1, nw), **kw)
        res = torch.empty((self.B, 4), **kw)
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_get_iterate(self._h, _ptr(x), _ptr(u), _ptr(pi), _ptr(bnd), _ptr(res), self._stream()),
                        "mpcrl_get_iterate")
        return x, u, pi, bnd, res

    def set_iterate(self, x, u, pi, bnd=None) -> None:
        """bnd = None: the bound multipliers / slacks are reset to the cold state and the next solve starts its interior point from
        the default point (MPCRL_COLD_DUAL) — an initial guess for x, u, pi, not a warm start."""
        nw = self.nx + self.nu
        x = self._dev(x, (self.B, self.N + 1, self.nx))
        u = self._dev(u, (self.B, self.N, self.nu))
        pi = self._dev(pi, (self.B, self.N, self.nx))
        bnd = None if bnd is None else self._dev(bnd, (self.B, 10, self.N + 1, nw))
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_set_iterate(self._h, _ptr(x), _ptr(u), _ptr(pi), _ptr(bnd), self._stream()),
                        "mpcrl_set_iterate")
        self.has_iterate, self.duals_valid = True, bnd is not None

    def _row_tables(self, x, u, pi, bnd, index):
        nw = self.nx + self.nu
        lens = ((self.N + 1) * self.nx, self.N * self.nu, self.N * self.nx, 10 * (self.N + 1) * nw)
        for t, n in zip((x, u, pi, bnd), lens):
            if t is not None and not (t.dtype == torch.float64 and t.is_contiguous() and t.device == self.device and t.numel() % n == 0):
                raise ValueError("iterate tables are contiguous float64 tensors on the handle's device with whole rows")
        if index is not None and not (index.dtype == torch.int64 and index.is_contiguous() and index.numel() == self.B and index.device == self.device):
            raise ValueError("index: contiguous int64 [B] on the handle's device")

    def get_iterate_rows(self, x, u, pi, bnd, index: Optional[torch.Tensor] = None) -> None:
        """Table row index[i] := stored iterate of instance i (mpcrl_get_iterate_rows; the tables are the caller's, any number of rows,
        rows laid out as get_iterate's; None = skip that array)."""
        self._row_tables(x, u, pi, bnd, index)
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_get_iterate_rows(self._h, _ptr(x), _ptr(u), _ptr(pi), _ptr(bnd), _ptr(index), self._stream()), "mpcrl_get_iterate_rows")

    def set_iterate_rows(self, x, u, pi, bnd, index: Optional[torch.Tensor] = None) -> None:
        """Stored iterate of instance i := table row index[i] (mpcrl_set_iterate_rows); bnd = None as in set_iterate."""
        self._row_tables(x, u, pi, bnd, index)
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_set_iterate_rows(self._h, _ptr(x), _ptr(u), _ptr(pi), _ptr(bnd), _ptr(index), self._stream()), "mpcrl_set_iterate_rows")
        self.has_iterate, self.duals_valid = True, bnd is not None

    def get_lagrangian(self) -> torch.Tensor:
        """[B] Lagrangian of the mirror NLP at the iterate of the last solve (nlp.L, nlp.py:1180,1390)."""
        L = torch.empty((self.B,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.mpcrl_get_lagrangian(self._h, _ptr(L), self._stream()), "mpcrl_get_lagrangian")
        return L

    def workspace_bytes(self) -> int:
        return int(self.lib.mpcrl_workspace_bytes(self._h))
