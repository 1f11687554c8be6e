import sys, numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, cartpole_ocp
from oracle.problems import make_cartpole
from oracle import cpu_port as port
port.build(); P = make_cartpole()
rng = np.random.default_rng(5)
for name, B, lohi, mi in (("bench", 4096, None, None), ("hard", 2048, np.array([2.0, 3.0, np.pi, 3.0]), 60)):
    if lohi is None:
        x0 = np.zeros((B, 4)); x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
    else:
        x0 = rng.uniform(-lohi, lohi, (B, 4))
    ocp = cartpole_ocp()
    mpc = MPCBatch(ocp, B)
    kw = {}
    if mi: mpc.set_options(max_iter=mi); kw = dict(max_iter=mi)
    r = mpc.solve(torch.as_tensor(x0, device='cuda'), sens_v=True, sens_pi=True, cold=True)
    ref = port.solve(P, x0, **kw)
    st = r.status.cpu().numpy(); it = r.iters.cpu().numpy()
    same = st == ref.status
    ok = (st == 0) & (ref.status == 0)
    def rel(a, b): return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0
    print(name, "B", B, "status equal %.4f" % same.mean(), "converged gpu %.4f oracle %.4f" % ((st == 0).mean(), (ref.status == 0).mean()),
          "sqp iters equal %.4f (max diff %d)" % ((it[ok, 0] == ref.sqp_iter[ok]).mean(), np.abs(it[ok, 0] - ref.sqp_iter[ok]).max()),
          "ipm iters equal %.4f" % ((it[ok, 1] == ref.ipm_iter[ok]).mean() if hasattr(ref, 'ipm_iter') else -1),
          "u0 rel %.2e V rel %.2e dV rel %.2e" % (rel(r.u0.cpu().numpy()[ok], ref.u0[ok]), rel(r.V.cpu().numpy()[ok], ref.V[ok]), rel(r.dV_dp.cpu().numpy()[ok], ref.dV[ok])))
