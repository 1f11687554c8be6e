import numpy as np, torch, time, sys
sys.path.insert(0,'/root/repo')
from mpc4rl_amd import MPCBatch, cartpole_ocp
B=4096
rng=np.random.default_rng(0)
x0=np.zeros((B,4)); x0[:,2]=rng.uniform(0.9*np.pi,1.1*np.pi,B)
mpc=MPCBatch(cartpole_ocp(),B)
def run(x):
    xt=torch.as_tensor(x,device='cuda')
    for _ in range(2): r=mpc.solve(xt,sens_v=True,sens_pi=True,cold=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): r=mpc.solve(xt,sens_v=True,sens_pi=True,cold=True)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/10*1e3, r
t,r=run(x0); print("random order ms",t)
it=r.iters.cpu().numpy()
t2,_=run(x0[np.argsort(x0[:,2])]); print("sorted by theta ms",t2)
t3,_=run(x0[np.argsort(it[:,1],kind='stable')]); print("sorted by ipm iters ms",t3)
