"""Cycle breakdown of the cartpole solve kernel by phase (per wavefront).  Needs a library built with -DMPCRL_PROFILE_PHASES:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -DMPCRL_PROFILE_PHASES -Iinclude mpc4rl_amd/csrc/mpcrl_api.hip -o mpc4rl_amd/libmpcrl_hip.so
"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp, _lib
B = 4096
ocp = cartpole_ocp(); mpc = MPCBatch(ocp, B)
rng = np.random.default_rng(0); x0 = np.zeros((B, 4)); x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
x0 = torch.as_tensor(x0, device='cuda')
lib = _lib.load(); out = (C.c_ulonglong * 16)()
mpc.solve(x0, cold=True); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
r = mpc.solve(x0, cold=True); torch.cuda.synchronize(); lib.mpcrl_debug_phases(out, 1)
names = ["0 residuals + stop test", "1 predictor barrier terms", "2 corrector: kff, d (stage lanes)", "3 forward chain (pred)",
         "4 predictor rows, mu_aff", "5 corrector barrier terms, c", "6 backward chain", "7 forward chain (corr)",
         "8 step length + update", "9 SQP: linearise + residuals", "10 SQP: step, tolerances", "11 QP setup", "12 factor: publish", "13 factor: MFMA sweep", "14 fetch Dx, Du (Dnu)", "15 (sliced: rotation)"]
waves = (B + 2) // 3
tot = sum(out[i] for i in range(16))
print("IPM iterations mean %.1f, SQP mean %.1f" % (r.iters[:, 1].float().mean().item(), r.iters[:, 0].float().mean().item()))
for i in range(16):
    print("%-40s %9.0f cycles/wave  %5.1f%%" % (names[i], out[i] / waves, 100.0 * out[i] / tot))
print("total %.0f cycles per wave (s_memtime ticks)" % (tot / waves))
