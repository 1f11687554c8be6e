import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from mpc4rl_amd import MPCBatch, cartpole_ocp
B=4096
rng=np.random.default_rng(0)
x0=np.zeros((B,4)); x0[:,2]=rng.uniform(0.9*np.pi,1.1*np.pi,B)
mpc=MPCBatch(cartpole_ocp(),B); mpc.set_variant(1)
r=mpc.solve(x0,cold=True); torch.cuda.synchronize()
res=mpc.get_iterate()[4].cpu().numpy()
print("sweep cycles, total cycles, ipm iterations:", res[-1][:3], "frac", res[-1][0]/res[-1][1], "per iter sweep", res[-1][0]/res[-1][2], "per iter total", res[-1][1]/res[-1][2])
