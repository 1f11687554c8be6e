"""Status codes of instances with a non-finite x0 (NaN, overflow) on the three models: expected 1 (or 4 where the QP breaks first)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp, linear_system_ocp, chain_mass_ocp

for name, ocp, nx, B in (("cartpole", cartpole_ocp(), 4, 131), ("linear", linear_system_ocp(), 2, 131)):
    x0 = np.random.default_rng(4).uniform(-0.3, 0.3, (B, nx)); x0[7] = np.nan; x0[9] = 1e30
    r = MPCBatch(ocp, B).solve(x0, sens_v=True, sens_pi=True, cold=True)
    print(name, r.status[:12].tolist(), r.iters[:12, 0].tolist(), r.V[7].item(), r.dV_dp[7, :4].tolist(), r.dV_dp[9, :4].tolist())
ocp = chain_mass_ocp(n_mass=3)
m = MPCBatch(ocp, 8)
x0 = np.tile(np.asarray(ocp.x0_default if hasattr(ocp, "x0_default") else np.zeros(m.nx)), (8, 1)).astype(float)
x0[3, 0] = np.nan
r = m.solve(x0, sens_v=True, cold=True)
print("chain3", r.status.tolist(), r.iters[:, 0].tolist())
