"""Cost of the empty SQP rounds of the chain solve: (max_iter + 1) x (point, lin, qp) launches are queued per solve and instances that
have finished return at once; with max_iter = 50 about 44 of the 51 rounds find nothing to do.  Same batch, max_iter 50 vs 12."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, chain_mass_ocp
B = 1024
for mi in (50, 12):
    ocp = chain_mass_ocp(max_iter=mi)
    x0 = np.tile(ocp.x0, (B, 1)); x0[:, 12:] += np.random.default_rng(0).normal(0, 1e-2, (B, 9))
    mpc = MPCBatch(ocp, B); xt = torch.as_tensor(x0, device='cuda')
    for _ in range(2): r = mpc.solve(xt, sens_v=True, sens_pi=True, cold=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): r = mpc.solve(xt, sens_v=True, sens_pi=True, cold=True)
    torch.cuda.synchronize()
    print("max_iter %d: %.2f ms per %d solves, converged %.3f, SQP it max %d" % (mi, (time.perf_counter() - t) / 5 * 1e3, B, float((r.status == 0).float().mean()), int(r.iters[:, 0].max())))
