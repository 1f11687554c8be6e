"""Does balancing the time-sliced wavefronts pay?  The four instances of a sliced wavefront (positions w, W + w, 2W + w, 3W + w) share
its lifetime; with the batch sorted along the x0 coordinate of largest spread and dealt out boustrophedon (ranks w, 2W-1-w, 2W+w,
4W-1-w) the sums of their iteration counts are nearly equal.  Host-built permutation through mpcrl_set_order, reorder=False."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from mpc4rl_amd import MPCBatch, cartpole_ocp
from mpc4rl_amd.batch import _ptr

B = 4096
x0 = torch.as_tensor(bench.make_inputs(B, 0), device="cuda")
mpc = MPCBatch(cartpole_ocp(), B)
W = (B + 3) // 4


def perm_serp(key):
    o = np.argsort(key, kind="stable")
    p = np.empty(B, np.int64)
    for q in range(4):
        seg = o[q * W:(q + 1) * W]
        p[q * W:q * W + len(seg)] = seg if q % 2 == 0 else seg[::-1]
    return torch.as_tensor(p, dtype=torch.int32, device="cuda")


def timeit(perm, n=40):
    mpc._check(mpc.lib.mpcrl_set_order(mpc._h, _ptr(perm), mpc._stream()), "set_order")
    for _ in range(5):
        r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True, reorder=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True, reorder=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r


th = x0[:, 2].cpu().numpy()
t0, r0 = timeit(None)
t1, r1 = timeit(perm_serp(th))
t2, r2 = timeit(torch.as_tensor(np.argsort(th), dtype=torch.int32, device="cuda"))
rng = np.random.default_rng(1)
t3, r3 = timeit(torch.as_tensor(rng.permutation(B), dtype=torch.int32, device="cuda"))
print(f"identity {t0:.4f} ms, boustrophedon by theta {t1:.4f} ms, sorted {t2:.4f} ms, random {t3:.4f} ms")
for r in (r1, r2, r3):
    assert torch.equal(r.u0, r0.u0) and torch.equal(r.dV_dp, r0.dV_dp) and torch.equal(r.iters, r0.iters)
print("bit-identical results")
