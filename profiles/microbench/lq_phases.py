"""Phase breakdown of lq_solve_kernel (linear_kernel.hpp): shader-clock ticks of lane 0 per phase, summed over the wavefronts.
  make -C mpc4rl_amd/csrc -j8 OUT=../../ab/prof.so BUILD=build_prof EXTRA=-DMPCRL_PROFILE_PHASES
  MPCRL_LIB_PATH=ab/prof.so python profiles/microbench/lq_phases.py
"""
import ctypes as C, numpy as np, torch, sys, time
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, linear_system_ocp, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
x0 = np.column_stack([rng.uniform(0.15, 0.85, B), rng.uniform(-0.5, 0.5, B)])
mpc = MPCBatch(linear_system_ocp(discount_factor=0.99), B); x0t = torch.as_tensor(x0, device='cuda')
lib = _lib.load()
out = (C.c_ulonglong * 16)()
mpc.solve(x0t, cold=True); torch.cuda.synchronize()
lib.mpcrl_debug_phases(out, 1)
t = time.perf_counter(); r = mpc.solve(x0t, cold=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
lib.mpcrl_debug_phases(out, 1)
names = ["0 resid+check", "1 barrier pred", "2 factor", "3 fwd pred", "4 rows pred", "5 barrier corr", "6 bwd_vec", "7 fwd corr", "8 rows corr+step",
         "9 linearise+res", "10 qp setup", "11", "12", "13", "14", "15"]
tot = sum(out[i] for i in range(15))      # bucket 15 is the wavefront's life on the constant 100 MHz clock, not shader cycles
it = r.iters.cpu().numpy()
W = (B + 3) // 4
print("solve %.3f ms, ipm mean %.2f max %d; wavefronts %d" % (dt*1e3, it[:, 1].mean(), it[:, 1].max(), W))
for i in range(15):
    if out[i]:
        print("%-20s %9.0f ticks per wavefront  %5.1f%%" % (names[i], out[i] / W, 100.0 * out[i] / max(tot, 1)))
print("total ticks per wavefront:", tot / W)
print("wavefront life (s_memrealtime, 100 MHz): %.3f ms" % (out[15] / W * 1e-5))
