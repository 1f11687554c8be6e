"""Cycle breakdown of small_sens_kernel<CartpoleDev> by phase (per wavefront of three instances): the phase counters of a solve + sensitivity
call minus those of a solve-only call.  Needs the -DMPCRL_PROFILE_PHASES library (MPCRL_LIB_PATH=ab/prof.so)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc4rl_amd import MPCBatch, cartpole_ocp, _lib
B = 4096
mpc = MPCBatch(cartpole_ocp(), B)
rng = np.random.default_rng(0); x0 = np.zeros((B, 4)); x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
x0 = torch.as_tensor(x0, device='cuda')
lib = _lib.load()
def run(**kw):
    out = (C.c_ulonglong * 16)()
    mpc.solve(x0, cold=True, **kw); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
    mpc.solve(x0, cold=True, **kw); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
    return np.array([out[i] for i in range(16)], float)
a, b = run(), run(sens_v=True, sens_pi=True)
d = b - a
names = {9: "loads + linearisation with second-order jets", 12: "Hessian / barrier set-up + publish", 13: "adjoint factor sweep (MFMA)", 3: "forward chain",
         14: "fetch Dx, Du", 8: "Dnu, mixed-term jets, reductions, stores"}
waves = (B + 2) // 3
tot = sum(d[i] for i in names)
for i, n in names.items():
    print("%-48s %8.0f cycles/wave  %5.1f%%" % (n, d[i] / waves, 100 * d[i] / tot))
print("total %.0f cycles per wavefront (%.1f us at 2.4 GHz); other slots changed by %s" % (tot / waves, tot / waves / 2400, [int(d[i] / waves) for i in range(16) if i not in names]))
