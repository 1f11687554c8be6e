"""Cold solves of closed-loop states (what the TD3 target actor and the policy-gradient solve see: replay samples along swing-up
trajectories, each instance needing a different number of SQP iterations): time-sliced vs plain launch of the small solve kernel,
with and without the opt-in divergence exit.    MPCRL_TIME_SLICE=0|1 python profiles/microbench/replay_cold_solve.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp, BatchedCartPoleSwingUpEnv
B = 4096
env = BatchedCartPoleSwingUpEnv(B, device="cuda", seed=0)
obs = env.reset()
mpc = MPCBatch(cartpole_ocp(), B)
r = mpc.solve(obs, cold=True)
states = []
for step in range(40):
    a = (2.0 * ((r.u0 + 30.0) / 60.0) - 1.0 + 0.1 * torch.randn_like(r.u0)).clamp(-1, 1)
    obs, _, term, trunc = env.step(a)
    r = mpc.solve(obs)
    if step % 4 == 3: states.append(obs.clone())
pool = torch.cat(states)
idx = torch.randint(0, pool.shape[0], (B,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
x = pool[idx].contiguous()
for rule in (None, (10, 0.1)):
    cold = MPCBatch(cartpole_ocp(), B)
    if rule: cold.set_exit_rule(*rule)
    for sens in (False, True):
        for _ in range(4): cold.solve(x, cold=True, sens_pi=sens); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): rr = cold.solve(x, cold=True, sens_pi=sens)
        torch.cuda.synchronize()
        it = rr.iters.cpu().numpy()
        if not sens: print("   launch tuner (time-sliced ms, plain ms, preferred):", cold.launch_times())
        print("MPCRL_TIME_SLICE=%s exit rule %s sens_pi %s: %.3f ms per %d cold solves; converged %.3f, SQP it mean %.1f max %d, IPM it mean %.1f" % (
            os.environ.get("MPCRL_TIME_SLICE", "auto"), rule, sens, (time.perf_counter() - t) / 20 * 1e3, B, float((rr.status == 0).float().mean()), it[:, 0].mean(), it[:, 0].max(), it[:, 1].mean()))
