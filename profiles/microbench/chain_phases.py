"""Phase breakdown of the chain kernel (one wavefront per instance): shader-clock ticks of lane 0 per phase, summed over the
batch.  Needs a library built with -DMPCRL_PROFILE_PHASES, selected through MPCRL_LIB_PATH:
  make -C mpc4rl_amd/csrc -j8 OUT=../../ab/prof.so BUILD=build_prof EXTRA=-DMPCRL_PROFILE_PHASES
  MPCRL_LIB_PATH=ab/prof.so python profiles/microbench/chain_phases.py [n_mass]
"""
import ctypes as C, numpy as np, torch, sys, time
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, chain_mass_ocp, _lib
n_mass = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ocp = chain_mass_ocp(n_mass=n_mass); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(0)
M = n_mass - 2
x0 = np.tile(ocp.x0, (B, 1))
x0[:, 3*(M+1):] += rng.normal(0, 1e-2, (B, 3*M))
mpc = MPCBatch(ocp, B); x0t = torch.as_tensor(x0, device='cuda')
lib = _lib.load()
out = (C.c_ulonglong * 16)()
mpc.solve(x0t, cold=True); torch.cuda.synchronize()
lib.mpcrl_debug_phases(out, 1)
t = time.perf_counter(); r = mpc.solve(x0t, cold=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
lib.mpcrl_debug_phases(out, 1)
names = ["0 resid", "1 barrier", "2 factor (= 10..13)", "3 bwd_vec", "4 forward", "5 steplen", "6 lin: point pass", "7 update+qp setup", "8 lin: direction pass",
         "9 cost+nlp res", "10 fac:T", "11 fac:M", "12 fac:chol/K", "13 fac:P", "14 sqp step", "15"]
tot = sum(out[i] for i in range(16) if i != 2)   # bucket 2 is stamped by the caller of the factor sweep: it repeats 10..13
it = r.iters.cpu().numpy()
print("n_mass %d: solve %.2f ms, sqp mean %.2f, ipm mean %.2f" % (n_mass, dt*1e3, it[:, 0].mean(), it[:, 1].mean()))
for i in range(16):
    if out[i]:
        print("%-20s %9.1f us per instance  %5.1f%%" % (names[i], out[i] / B / 2400.0, 100.0 * out[i] / max(tot, 1)))
print("total us per instance (at 2.4 GHz):", tot / B / 2400.0)
