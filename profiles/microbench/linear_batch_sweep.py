"""Linear-system solve + sensitivities per batch size: lq_solve_kernel (default) against the one-stage-per-lane kernels (MPCRL_LINEAR_SPL=1)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, linear_system_ocp
def run(B, spl):
    os.environ['MPCRL_LINEAR_SPL'] = spl
    rng = np.random.default_rng(0)
    x0 = torch.as_tensor(np.column_stack([rng.uniform(0.15, 0.85, B), rng.uniform(-0.5, 0.5, B)]), device='cuda')
    m = MPCBatch(linear_system_ocp(discount_factor=0.99), B)
    for _ in range(5): m.solve(x0, sens_v=True, sens_pi=True, cold=True)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): m.solve(x0, sens_v=True, sens_pi=True, cold=True)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 30)
    return best
print("%8s %22s %22s" % ("batch", "three stages per lane", "one stage per lane"))
for B in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    a, b = run(B, '3'), run(B, '1')
    print("%8d %9.3f ms %7.2f M/s %9.3f ms %7.2f M/s" % (B, a * 1e3, B / a / 1e6, b * 1e3, B / b / 1e6), flush=True)
