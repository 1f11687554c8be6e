import numpy as np, torch, time, sys
sys.path.insert(0,'/root/repo')
from mpc4rl_amd import MPCBatch, chain_mass_ocp
B=1024
ocp=chain_mass_ocp()
rng=np.random.default_rng(0)
x0=np.tile(ocp.x0,(B,1)); x0[:,12:]+=rng.normal(0,1e-2,(B,9))
x0t=torch.as_tensor(x0,device='cuda')
mpc=MPCBatch(ocp,B)
def t(f,n=3):
    f(); torch.cuda.synchronize(); s=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-s)/n*1e3, r
for mi in (0,1,2,50):
    mpc.set_options(max_iter=mi)
    ms,r=t(lambda: mpc.solve(x0t,cold=True,reorder=False))
    it=r.iters.cpu().numpy()
    print("max_iter",mi,"ms %.2f"%ms,"sqp",it[:,0].mean(),"ipm",it[:,1].mean())
mpc.set_options(max_iter=50)
ms,r=t(lambda: mpc.solve(x0t,cold=True,sens_v=True,reorder=False)); print("solve+sensV ms %.2f"%ms)
ms,r=t(lambda: mpc.solve(x0t,cold=True,sens_v=True,sens_pi=True,reorder=False)); print("solve+sensV+PI ms %.2f"%ms)
