"""ctypes binding of the C++ CPU oracle port (oracle/cpu).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .problems import Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libmpc_oracle.so")
# MPC_ORACLE_LIB: another build of the same source (tests/test_oracle_sanitizers.py runs the ASan + UBSan one, `make asan`)
_LIB_OVERRIDE = os.environ.get("MPC_ORACLE_LIB")

SENS_V, SENS_PI, WARM, EXACT, RTI = 1, 2, 4, 8, 16
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OracleSpec(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("N", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("np", C.c_int), ("cost_kind", C.c_int),
        ("dT", C.c_double), ("gamma", C.c_double), ("h", C.c_double), ("rk_steps", C.c_int), ("tol", C.c_double),
        ("max_iter", C.c_int),
        ("lb0", _dp), ("ub0", _dp), ("lb", _dp), ("ub", _dp), ("lbe", _dp), ("ube", _dp), ("soft", _ip), ("zl", _dp), ("zu", _dp),
        ("consts", _dp), ("n_consts", C.c_int), ("exit_window", C.c_int), ("exit_factor", C.c_double),
    ]


def build(force: bool = False) -> str:
    if _LIB_OVERRIDE:
        return _LIB_OVERRIDE
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "cpu")] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.mpc_oracle_solve.restype = C.c_int
    return _lib


def stage_bounds(P: Problem):
    """Box bounds in stage-vector order v = [u; x]; +-1e30 = absent."""
    nw = P.nu + P.nx
    lb, ub = np.full(nw, -1e30), np.full(nw, 1e30)
    lb[: P.nu], ub[: P.nu] = P.lbu, P.ubu
    soft = np.zeros(nw, np.int32)
    zl, zu = np.zeros(nw), np.zeros(nw)
    for j, ix in enumerate(P.idxbx):
        lb[P.nu + ix], ub[P.nu + ix] = P.lbx[j], P.ubx[j]
    for n, j in enumerate(P.idxsbx):
        ix = P.idxbx[j]
        soft[P.nu + ix] = 1
        zl[P.nu + ix], zu[P.nu + ix] = P.zl[n], P.zu[n]
    lbe, ube = np.full(P.nx, -1e30), np.full(P.nx, 1e30)
    for j, ix in enumerate(P.idxbx_e):
        lbe[ix], ube[ix] = P.lbx_e[j], P.ubx_e[j]
    return lb, ub, lbe, ube, soft, zl, zu


def model_consts(P: Problem) -> tuple:
    """(model id, h, rk_steps, consts) — the per-model constant blob read by mpc_oracle.cpp."""
    if P.name == "cartpole":
        return 0, P.extra["h"], 1, np.zeros(1)     # the cost block is read from p (models.hpp Cartpole)
    if P.name == "linear_system":
        return 1, 0.0, 0, P.extra["P"].reshape(-1).copy()
    if P.name.startswith("chain_mass"):
        return 2, P.dT / 2, 2, P.extra["x_ss"].copy()
    raise KeyError(P.name)


@dataclass
class PortResult:
    status: np.ndarray
    sqp_iter: np.ndarray
    ipm_iter: np.ndarray
    res: np.ndarray
    X: np.ndarray
    U: np.ndarray
    PI: np.ndarray
    BND: np.ndarray       # (B, 10, N+1, nu+nx): lam_l, lam_u, t_l, t_u, s_l, s_u, lam_sl, lam_su, t_sl, t_su
    u0: np.ndarray
    V: np.ndarray
    dV: Optional[np.ndarray]
    dpi: Optional[np.ndarray]


def solve(P: Problem, x0, p=None, u0fix=None, gamma=None, flags=SENS_V | SENS_PI, warm: Optional[PortResult] = None,
          max_iter=None, tol=None, nthreads=0, want_bnd=True, exact=False, rti=False, exit_window=0, exit_factor=0.1) -> PortResult:
    """exact=True: the frozen exact-QP mode (ORACLE_EXACT, oracle/cpu/mpc_oracle.h); rti=True: one SQP iteration from ``warm``;
    exit_window > 0: the product's opt-in divergence exit (mpcrl_set_exit_rule: every exit_window SQP iterations the best residual
    must have dropped below exit_factor x its value at the previous check, else status 2)."""
    x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, float)))
    B = x0.shape[0]
    nw = P.nu + P.nx
    pp = np.ascontiguousarray(P.p0 if p is None else np.asarray(p, float))
    per = int(pp.ndim == 2)
    assert pp.shape[-1] == P.n_p and (not per or pp.shape[0] == B)
    lb, ub, lbe, ube, soft, zl, zu = stage_bounds(P)
    lb0, ub0 = lb[: P.nu].copy(), ub[: P.nu].copy()
    mid, h, rk, consts = model_consts(P)
    consts = np.ascontiguousarray(consts, float)
    keep = [lb, ub, lbe, ube, soft, zl, zu, lb0, ub0, consts]

    def dptr(a):
        return a.ctypes.data_as(_dp)
    sp = OracleSpec(
        model=mid, N=P.N, nx=P.nx, nu=P.nu, np=P.n_p, cost_kind=0 if P.cost_kind == "NLS" else 1, dT=P.dT,
        gamma=P.gamma if gamma is None else gamma, h=h, rk_steps=rk, tol=P.tol if tol is None else tol,
        max_iter=P.max_iter if max_iter is None else max_iter,
        lb0=dptr(lb0), ub0=dptr(ub0), lb=dptr(lb), ub=dptr(ub), lbe=dptr(lbe), ube=dptr(ube), soft=soft.ctypes.data_as(_ip),
        zl=dptr(zl), zu=dptr(zu), consts=dptr(consts), n_consts=len(consts), exit_window=int(exit_window), exit_factor=float(exit_factor))
    flags |= (EXACT if exact else 0) | (RTI if rti else 0)
    if warm is not None:
        flags |= WARM
        X, U, PI, BND = warm.X.copy(), warm.U.copy(), warm.PI.copy(), warm.BND.copy()
    else:
        X, U, PI = np.zeros((B, P.N + 1, P.nx)), np.zeros((B, P.N, P.nu)), np.zeros((B, P.N, P.nx))
        BND = np.zeros((B, 10, P.N + 1, nw)) if want_bnd else None
    u0 = np.zeros((B, P.nu))
    V = np.zeros(B)
    dV = np.zeros((B, P.n_p)) if flags & SENS_V else None
    dpi = np.zeros((B, P.nu, P.n_p)) if flags & SENS_PI else None
    status, nsqp, nipm = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    res = np.zeros((B, 4))
    uf = None
    if u0fix is not None:
        uf = np.ascontiguousarray(np.atleast_2d(np.asarray(u0fix, float)))
        assert uf.shape == (B, P.nu)
    rc = lib().mpc_oracle_solve(
        C.byref(sp), B, dptr(x0), dptr(uf) if uf is not None else None, dptr(pp), per, flags, dptr(X), dptr(U), dptr(PI), dptr(BND) if BND is not None else None,
        dptr(u0), dptr(V), dptr(dV) if dV is not None else None, dptr(dpi) if dpi is not None else None,
        status.ctypes.data_as(_ip), nsqp.ctypes.data_as(_ip), nipm.ctypes.data_as(_ip), dptr(res), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"mpc_oracle_solve returned {rc}")
    del keep
    return PortResult(status, nsqp, nipm, res, X, U, PI, BND, u0, V, dV, dpi)
