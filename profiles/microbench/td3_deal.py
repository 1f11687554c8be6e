"""Replay-state cold solves: does dealing the batch out by a difficulty key balance the time-sliced wavefronts?"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp
from mpc4rl_amd.td3 import BatchedTD3
from mpc4rl_amd.envs import BatchedCartPoleSwingUpEnv
dev = torch.device('cuda:0')
E = 4096
env = BatchedCartPoleSwingUpEnv(E, device=dev, seed=0)
agent = BatchedTD3(cartpole_ocp(), env, batch_size=E, buffer_steps=64, policy_delay=2, lr_actor=1e-6, seed=0, device=dev)
for _ in range(70):
    agent.collect(1)
obs, nxt, act, rew, done = agent.buffer.sample(E, agent.gen)
x = nxt.to(torch.float64).contiguous()
mpc = MPCBatch(cartpole_ocp(), E, device=dev)
def tm(perm):
    import ctypes as C
    def f():
        if perm is None:
            r = mpc.solve(x, cold=True)
        else:
            mpc._check(mpc.lib.mpcrl_set_order(mpc._h, C.c_void_p(perm.data_ptr()), mpc._stream()), "set_order")
            r = mpc.solve(x, cold=True, reorder=False)
        return r
    for _ in range(3): r = f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 20 * 1e3, r
t0, r0 = tm(None)
it = r0.iters[:, 0].double()
print("replay next-states: no order %.3f ms; SQP iterations mean %.2f max %d; IPM mean %.2f" % (t0, float(it.mean()), int(it.max()), float(r0.iters[:, 1].double().mean())))
th = x[:, 2]
keys = {"1-cos(theta)": 1 - torch.cos(th), "|theta_dot|": x[:, 3].abs(), "oracle: the iteration count itself": it + 1e-3 * r0.iters[:, 1].double()}
for name, k in keys.items():
    perm = torch.argsort(k).to(torch.int32).contiguous()
    t1, r1 = tm(perm)
    assert torch.equal(r1.status, r0.status)
    print("dealt by %-36s %.3f ms" % (name, t1))
