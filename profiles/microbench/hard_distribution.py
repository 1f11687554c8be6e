"""Lock-step tail: what a batch costs when part of it does not converge.  Full-step SQP without globalisation (what the reference
runs, config/cartpole.yaml:8-14) fails on ~15 % of initial states drawn from the whole state box; such an instance keeps its
wavefront iterating until max_iter (500 in the reference's yaml).  Times the bench distribution, the hard distribution at
max_iter = 500 without and with the opt-in divergence exit (mpcrl_set_exit_rule) and at max_iter = 50, and closed-loop warm solves
along a rollout.
    python profiles/microbench/hard_distribution.py
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp, BatchedCartPoleSwingUpEnv

def timed(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r

B = 4096
rng = np.random.default_rng(7)
xb = np.zeros((B, 4)); xb[:, 2] = np.random.default_rng(0).uniform(0.9 * np.pi, 1.1 * np.pi, B)
xh = rng.uniform(-1, 1, (B, 4)) * np.array([2.0, 3.0, np.pi, 5.0])
conv_off = None
for name, x0, mi, rule in (("bench distribution", xb, 500, None), ("bench distribution, exit rule (10, 0.1)", xb, 500, (10, 0.1)),
                           ("hard distribution (whole box)", xh, 500, None), ("hard distribution, exit rule (15, 0.1)", xh, 500, (15, 0.1)),
                           ("hard distribution, exit rule (10, 0.1)", xh, 500, (10, 0.1)), ("hard distribution, exit rule (6, 0.3)", xh, 500, (6, 0.3)),
                           ("hard distribution, max_iter 50", xh, 50, None)):
    mpc = MPCBatch(cartpole_ocp(max_iter=mi), B)
    if rule: mpc.set_exit_rule(*rule)
    xt = torch.as_tensor(x0, device="cuda")
    ms, r = timed(lambda: mpc.solve(xt, sens_v=True, sens_pi=True, cold=True))
    st = r.status.cpu().numpy(); it = r.iters.cpu().numpy()
    if name.startswith("hard distribution (whole"): conv_off = st == 0
    lost = "" if (conv_off is None or not name.startswith("hard") or rule is None) else "; lost %.1f %% of the rule-less run's converged" % (100.0 * (conv_off & (st != 0)).sum() / conv_off.sum())
    print("%-42s %8.2f ms per %d solves; converged %.3f, max-iter %.3f, QP-fail %.3f, NaN %.3f; SQP it mean %.1f max %d%s" % (
        name, ms, B, (st == 0).mean(), (st == 2).mean(), (st == 4).mean(), (st == 1).mean(), it[:, 0].mean(), it[:, 0].max(), lost))
env = BatchedCartPoleSwingUpEnv(B, device="cuda", seed=0)
obs = env.reset()
mpc = MPCBatch(cartpole_ocp(), B)
r = mpc.solve(obs, cold=True)
torch.cuda.synchronize()
ts, its = [], []
for step in range(60):
    a = 2.0 * ((r.u0 + 30.0) / 60.0) - 1.0
    obs, _, _, _ = env.step(a)
    torch.cuda.synchronize(); t = time.perf_counter()
    r = mpc.solve(obs)                    # warm-started full SQP from the previous solution, as mpc.get_action in a rollout
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3); its.append(r.iters.float().mean(0).cpu().numpy())
its = np.array(its)
print("closed loop, warm full SQP: %.3f ms per %d solves (median over 60 env steps), SQP it mean %.2f, IPM it mean %.2f, converged %.3f" % (
    np.median(ts), B, its[:, 0].mean(), its[:, 1].mean(), float((r.status == 0).float().mean())))
