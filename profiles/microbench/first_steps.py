import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mpc4rl_amd import MPCBatch, cartpole_ocp
dev = torch.device('cuda:0')
bench.measured_hbm_peak(dev)
B = 4096
mpc = MPCBatch(cartpole_ocp(), B, device=dev)
x0 = torch.as_tensor(bench.make_inputs(B, 0), device=dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
cpu = []
for i in range(40):
    t = time.perf_counter(); ev[i][0].record(); r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True); ev[i][1].record(); cpu.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
print("gpu ms per step:", [round(a.elapsed_time(b), 3) for a, b in ev])
print("cpu ms per call:", [round(c, 3) for c in cpu])
