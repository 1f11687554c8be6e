"""Phase breakdown of the chain (large) kernel's interior-point loop.  Needs a library built with -DMPCRL_PROFILE_PHASES:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPCRL_PROFILE_PHASES -Iinclude mpc4rl_amd/csrc/mpcrl_api.hip -o mpc4rl_amd/libmpcrl_hip.so
"""
import ctypes as C, numpy as np, torch, sys, time
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, chain_mass_ocp, _lib
ocp = chain_mass_ocp(n_mass=5); B = 1024
rng = np.random.default_rng(0)
x0 = np.tile(ocp.x0, (B, 1)); M = 3
x0[:, 3*(M+1):] += rng.normal(0, 1e-2, (B, 3*M))
mpc = MPCBatch(ocp, B); x0t = torch.as_tensor(x0, device='cuda')
lib = _lib.load()
out = (C.c_ulonglong * 16)()
mpc.solve(x0t, cold=True); torch.cuda.synchronize()
lib.mpcrl_debug_phases(out, 1)
t = time.perf_counter(); r = mpc.solve(x0t, cold=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
lib.mpcrl_debug_phases(out, 1)
names = ["0 resid", "1 barrier", "2 factor", "3 bwd_vec", "4 forward", "5 steplen", "6", "7 update"]
tot = sum(out[i] for i in range(8))
print("solve %.1f ms, ipm mean %.1f" % (dt*1e3, r.iters[:, 1].float().mean().item()))
for i in range(8):
    print("%-10s %8.2f us per instance-iteration-equivalent, %5.1f%%" % (names[i], out[i] / 100.0 / B / 22.9, 100.0 * out[i] / max(tot, 1)))
print("total ticks/inst (us):", tot / 100.0 / B)
