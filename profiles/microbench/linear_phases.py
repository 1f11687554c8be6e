"""Cycle breakdown of the linear-system solve kernel by phase (one instance per wavefront, N = 40).  Needs a library built with
-DMPCRL_PROFILE_PHASES (see cartpole_phases.py) and MPCRL_LIB_PATH pointing at it."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, linear_system_ocp, _lib
B = 4096
mpc = MPCBatch(linear_system_ocp(discount_factor=0.99), B)
rng = np.random.default_rng(0)
x0 = torch.as_tensor(np.column_stack([rng.uniform(0.15, 0.85, B), rng.uniform(-0.5, 0.5, B)]), device='cuda')
lib = _lib.load(); out = (C.c_ulonglong * 16)()
mpc.solve(x0, cold=True); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
r = mpc.solve(x0, cold=True); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
names = ["0 residuals + stop test", "1 predictor barrier terms", "2 backward sweep with factorisation (pred)", "3 forward sweep (pred)",
         "4 predictor rows, mu_aff", "5 corrector barrier terms, c", "6 backward sweep (corr)", "7 forward sweep (corr)",
         "8 step length + update", "9 SQP: linearise + residuals", "10 SQP: step, tolerances", "11 QP setup", "12 -", "13 -", "14 -", "15 -"]
tot = sum(out[i] for i in range(16))
print("IPM iterations mean %.1f, SQP mean %.1f" % (r.iters[:, 1].float().mean().item(), r.iters[:, 0].float().mean().item()))
for i in range(12):
    print("%-44s %9.0f cycles/wave  %5.1f%%" % (names[i], out[i] / B, 100.0 * out[i] / tot))
print("total %.0f cycles per wave (s_memtime ticks)" % (tot / B))
