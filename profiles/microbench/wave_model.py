"""Schedule model of the time-sliced cartpole launch: what the slowest of the 1024 wavefronts costs, from per-QP
interior-point traces of the CPU port.  Needs a SCRATCH copy of oracle/cpu/mpc_oracle.cpp + models.hpp in /tmp/ipmx with a trace
hook (dbg_set_trace: one byte per interior-point iteration, 1 = corrector ran, 2 = predictor step taken) and -D overrides of the
constants (K_CAP, K_WMAX, K_WC, K_AC, K_SKIP, K_MUF); kept here as the record of how the figures in DESIGN.md were produced, not as
part of the test suite.    python wave_model.py "-DK_CAP=3e-2" "-DK_CAP=5e-2 -DK_WMAX=1e-1" ..."""
import sys, os, numpy as np, ctypes as C, subprocess
sys.path.insert(0, '/root/repo')
from oracle import cpu_port
from oracle.problems import make_cartpole
import bench
def build(defs):
    so = '/tmp/ipmx/libw_%d.so' % (abs(hash(defs)) % 100000)
    subprocess.check_call("g++ -O3 -march=x86-64-v3 -std=c++17 -fPIC -fopenmp -shared %s /tmp/ipmx/mpc_oracle.cpp -o %s" % (defs, so), shell=True)
    return so
def run(defs, B=4096):
    lib = C.CDLL(build(defs)); lib.mpc_oracle_solve.restype = C.c_int
    cpu_port._lib = lib
    tr = np.zeros((B, 1024), np.uint8); lib.dbg_set_trace(tr.ctypes.data_as(C.POINTER(C.c_ubyte)))
    r = cpu_port.solve(make_cartpole(), bench.make_inputs(B, 0), nthreads=8, flags=0)
    lib.dbg_set_trace(None)
    return r, tr.reshape(B, 64, 16)
# cost model (units: % of a plain wavefront's life / count): predictor part of a tick, corrector part, step + update, SQP part, residual evaluation per QP
I_PRED, I_CORR, I_STEP, T_SQP, T_RES, ROT = 2.55, 0.94, 0.64, 2.0, 0.53, 0.35
def sim_sliced(tr, B):
    W = B // 4; tot = np.zeros(W)
    for w in range(W):
        inst = [w + j * W for j in range(4)]
        seq = [[tr[i, q][tr[i, q] > 0] for q in range(64) if tr[i, q, 0] > 0] for i in inst]
        pos = [0] * 4; slots = [0, 1, 2]; pk = 3; rr = 0; t = 0.0
        while True:
            act = [s for s in slots if pos[s] < len(seq[s])]
            if not act:
                break
            t += T_SQP + T_RES
            n = max(len(seq[s][pos[s]]) for s in act)
            for j in range(n):
                live = [s for s in act if len(seq[s][pos[s]]) > j]
                t += I_PRED + I_STEP
                if any(seq[s][pos[s]][j] == 1 for s in live): t += I_CORR
            for s in act: pos[s] += 1
            t += 0.3   # the final residual evaluation / convergence round is folded in T_SQP
            if pk is None: continue
            done = [i for i, s in enumerate(slots) if pos[s] >= len(seq[s])]
            t += ROT
            if done: slots[done[0]] = pk; pk = None
            else:
                v = rr % 3; rr += 1; slots[v], pk = pk, slots[v]
        tot[w] = t
    return tot
if __name__ == "__main__":
    for defs in sys.argv[1:]:
        r, tr = run(defs)
        tot = sim_sliced(tr, 4096)
        its = (tr > 0).sum((1, 2)); sk = (tr == 2).sum((1, 2))
        print("%-60s conv %.3f sqp %.2f ipm %.2f skipped %.2f | wave cost mean %.1f max %.1f" % (defs, (r.status == 0).mean(), r.sqp_iter.mean(), its.mean(), sk.mean(), tot.mean(), tot.max()))
