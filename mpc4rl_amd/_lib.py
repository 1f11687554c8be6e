"""ctypes binding of libmpcrl_hip.so (include/mpcrl.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPCRL_LIB_PATH") or os.path.join(_HERE, "libmpcrl_hip.so")   # the override is for instrumented builds (profiles/microbench)

ABI_VERSION = 130        # MPCRL_ABI_VERSION of include/mpcrl.h this binding was written against
SENS_V, SENS_PI, RTI, COLD = 1, 2, 4, 8
COLD_DUAL, NO_BND_STORE, EXACT_QP = 16, 32, 64        # (EXACT_QP: test-only, include/mpcrl.h)
MODEL_CARTPOLE, MODEL_LINEAR, MODEL_CHAIN = 0, 1, 2
COST_NLS, COST_EXTERNAL = 0, 1
NO_BOUND = 1e30
BOUNDS_U0, BOUNDS_STAGE, BOUNDS_TERMINAL = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class ProblemSpec(C.Structure):
    """MpcrlProblemSpec of include/mpcrl.h."""
    _fields_ = [
        ("model", C.c_int32), ("N", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("np", C.c_int32),
        ("cost_kind", C.c_int32), ("dT", C.c_double), ("gamma", C.c_double), ("h", C.c_double), ("rk_steps", C.c_int32),
        ("tol", C.c_double), ("max_iter", C.c_int32),
        ("lb0", _dp), ("ub0", _dp), ("lb", _dp), ("ub", _dp), ("lbe", _dp), ("ube", _dp),
        ("soft", _ip), ("zl", _dp), ("zu", _dp), ("consts", _dp), ("n_consts", C.c_int32),
    ]


EXPORTS = ["mpcrl_create", "mpcrl_destroy", "mpcrl_set_theta", "mpcrl_set_gamma", "mpcrl_set_options", "mpcrl_set_exit_rule", "mpcrl_set_order", "mpcrl_set_bounds", "mpcrl_set_cold_mask", "mpcrl_reset",
           "mpcrl_solve", "mpcrl_get_iterate", "mpcrl_set_iterate", "mpcrl_get_iterate_rows", "mpcrl_set_iterate_rows", "mpcrl_get_lagrangian", "mpcrl_weighted_grad_sum", "mpcrl_env_cartpole_step", "mpcrl_env_cartpole_reset", "mpcrl_env_linear_step", "mpcrl_policy_action", "mpcrl_critic_workspace_bytes", "mpcrl_critic_td_grad", "mpcrl_critic_dq_da", "mpcrl_replay_sample", "mpcrl_dpg_workspace_bytes", "mpcrl_dpg_grad", "mpcrl_td3_cartpole_collect", "mpcrl_td3_policy_post", "mpcrl_auto_order", "mpcrl_query_time_sliced", "mpcrl_set_launch_mode", "mpcrl_get_launch_times", "mpcrl_workspace_bytes", "mpcrl_version"]

_lib = None


def load():
    """Returns the loaded library; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"mpc4rl_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.mpcrl_version.restype = C.c_int
    if lib.mpcrl_version() != ABI_VERSION and not os.environ.get("MPCRL_SKIP_ABI_CHECK"):   # (the override: A/B runs against older builds) extern "C" links whatever the signatures are: refuse a library built from another header
        raise RuntimeError(f"mpc4rl_amd: {LIB_PATH} reports ABI version {lib.mpcrl_version()}, this binding expects {ABI_VERSION}; rebuild it "
                           "(`python -c 'import __graft_entry__ as g; g.build()'`).")
    vp = C.c_void_p
    lib.mpcrl_create.argtypes = [C.POINTER(ProblemSpec), C.c_int, C.c_int, C.POINTER(vp)]
    lib.mpcrl_destroy.argtypes = [vp]
    lib.mpcrl_set_theta.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.mpcrl_set_gamma.argtypes = [vp, C.c_double]
    lib.mpcrl_set_options.argtypes = [vp, C.c_double, C.c_int]
    lib.mpcrl_set_exit_rule.argtypes = [vp, C.c_int, C.c_double]
    lib.mpcrl_set_order.argtypes = [vp, vp, vp]
    lib.mpcrl_set_bounds.argtypes = [vp, C.c_int, _dp, _dp]
    lib.mpcrl_set_cold_mask.argtypes = [vp, vp, vp]
    lib.mpcrl_reset.argtypes = [vp, vp, vp]
    lib.mpcrl_solve.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_get_iterate.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_set_iterate.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mpcrl_get_iterate_rows.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_set_iterate_rows.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_get_lagrangian.argtypes = [vp, vp, vp]
    lib.mpcrl_auto_order.argtypes = [vp, vp, vp]
    lib.mpcrl_query_time_sliced.argtypes = [vp, C.c_int, vp]
    lib.mpcrl_set_launch_mode.argtypes = [vp, C.c_int]
    lib.mpcrl_get_launch_times.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mpcrl_weighted_grad_sum.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp]
    lib.mpcrl_env_cartpole_step.argtypes = [_dp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.mpcrl_env_cartpole_reset.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.mpcrl_env_linear_step.argtypes = [_dp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.mpcrl_policy_action.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp, vp]
    lib.mpcrl_critic_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mpcrl_critic_workspace_bytes.restype = C.c_int64
    lib.mpcrl_critic_td_grad.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, vp]
    lib.mpcrl_critic_dq_da.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_replay_sample.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_dpg_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.mpcrl_dpg_workspace_bytes.restype = C.c_int64
    lib.mpcrl_dpg_grad.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]
    lib.mpcrl_td3_cartpole_collect.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_double, vp, vp, vp, C.c_int,
                                               C.c_double, vp, vp, vp, vp, vp, vp]
    lib.mpcrl_td3_policy_post.argtypes = [vp, C.c_int, C.c_double, vp, C.c_double, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.mpcrl_workspace_bytes.argtypes = [vp]
    lib.mpcrl_workspace_bytes.restype = C.c_int64
    for name in EXPORTS:
        if name not in ("mpcrl_workspace_bytes", "mpcrl_critic_workspace_bytes", "mpcrl_dpg_workspace_bytes"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
