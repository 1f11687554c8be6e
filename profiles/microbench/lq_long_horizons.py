"""usage (GPU box): python profiles/microbench/lq_long_horizons.py — linear-system solve + sensitivities at horizons beyond a DPP row of three-stage
lanes (N >= 48).  The committed r06_lq_long_horizons.txt was taken with a build that still had lq_solve_kernel<3, 0> (packed segments,
__shfl: "spl 3" there) next to lq_solve_kernel<4, 16> (four stages per lane, rows: "spl 4"); the shipped library now runs <4, 16> at those
horizons under the default MPCRL_LINEAR_SPL=3, so this script compares the default with the one-stage-per-lane kernels
(MPCRL_LINEAR_SPL=1); 4096 instances, cold every call, ms per call (HIP events, median of 5 x 20)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc4rl_amd import MPCBatch, linear_system_ocp  # noqa: E402

B = 4096
rng = np.random.default_rng(0)
x0 = torch.as_tensor(np.column_stack([rng.uniform(0.15, 0.85, B), rng.uniform(-0.5, 0.5, B)]), device="cuda")
for N in (40, 47, 48, 55, 63):
    row = []
    ref = None
    for spl in ("1", "3"):
        os.environ["MPCRL_LINEAR_SPL"] = spl
        mpc = MPCBatch(linear_system_ocp(N=N, discount_factor=0.99), B)
        for _ in range(5):
            r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20)
        if ref is None:
            ref = r
        dv = float((r.dV_dp - ref.dV_dp).abs().max())
        row.append("spl %s: %.3f ms (ipm %.2f, status0 %.3f, |dV - spl1| %.1e)" % (spl, float(np.median(ts)), float(r.iters[:, 1].double().mean()), float((r.status == 0).double().mean()), dv))
        del mpc
    print("N = %d:  " % N + "   ".join(row), flush=True)
