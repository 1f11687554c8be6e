"""usage (GPU box): python profiles/microbench/fuzz_parity.py [seed] — the HIP path against the oracle's C++ port over shapes the suite does not
pin: random horizons, batch sizes (odd, below and above a wavefront's packing), per-instance parameters, discount factors, cold / warm /
cold-mask / Q-mode / RTI call sequences, all three OCP families.  Prints one line per configuration; exits non-zero on a mismatch
(status, SQP iteration count beyond +-1, u0*, V, dV/dp at 1e-6, du0*/dp at 1e-5 on instances both solve)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc4rl_amd import MPCBatch, cartpole_ocp, chain_mass_ocp, linear_system_ocp  # noqa: E402
from oracle import cpu_port  # noqa: E402
from oracle.problems import make_cartpole, make_chain_mass, make_linear_system  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = 0
DUMP = {}


def rel(a, b):
    if len(b) == 0:
        return np.zeros(0)
    a, b = np.asarray(a, float).reshape(len(b), -1), np.asarray(b, float).reshape(len(b), -1)
    nn = np.isnan(a) & np.isnan(b)
    a, b = np.where(nn, 0, a), np.where(nn, 0, b)
    return (np.abs(a - b) / np.maximum(np.abs(b).max(1, keepdims=True), 1.0)).max(1)


def check(tag, r, ref, dpi_tol=1e-5, unique=None, rti=False, among=None):
    """unique: for the convex linear-system QP — the port's solution of the same problem in its exact-QP mode from a cold start (ONE
    KKT point whatever the path): where the port's tuned iteration fails on a warm call (status 4: its re-evaluated residuals stall at
    their rounding level, or its interior point cycles on a degenerate QP — this box's rounding decides) or stops one interior-point
    iteration apart, the product is held to that point instead: u0, V at 1e-6, dV/dp at the QP tolerance 1e-4."""
    global bad
    st = r.status.cpu().numpy()
    it = r.iters.cpu().numpy()
    both = (st == 0) & (ref.status == 0)
    if among is not None:      # a chained call: only the instances both sides have solved so far start it from the same iterate
        import dataclasses
        ref = cpu_port.PortResult(*[getattr(ref, f.name)[among] if getattr(ref, f.name) is not None else None for f in dataclasses.fields(cpu_port.PortResult)])
        class _Q: pass
        rq = _Q()
        kt = torch.as_tensor(among, device=r.status.device)
        for n in ("u0", "V", "status", "iters", "dV_dp", "dpi_dp"):
            v = getattr(r, n)
            setattr(rq, n, v[kt] if v is not None else None)
        r, st, it = rq, st[among], it[among]
        if unique is not None:
            unique = cpu_port.PortResult(*[getattr(unique, f.name)[among] if getattr(unique, f.name) is not None else None for f in dataclasses.fields(cpu_port.PortResult)])
        if not among.any():
            return both
    if unique is not None:
        apart = (ref.status != 0) | (it[:, 1] != ref.ipm_iter)
        if apart.any() and (unique.status[apart] == 0).all():
            g_ok = st[apart] == 0
            eu = rel(r.u0.cpu().numpy()[apart], unique.u0[apart]).max()
            eV = rel(r.V.cpu().numpy()[apart], unique.V[apart]).max()
            edV = rel(r.dV_dp.cpu().numpy()[apart], unique.dV[apart]).max() if r.dV_dp is not None else 0.0
            f2 = (not g_ok.all()) or eu > 1e-6 or eV > 1e-6 or edV > 1e-4
            print("%-4s %s: %d instances where the port failed (%d) or stopped an interior-point iteration apart, held to the unique QP solution: u0 %.1e V %.1e dV %.1e, product converged on %d of them" % (
                "FAIL" if f2 else "ok", tag, apart.sum(), (ref.status != 0).sum(), eu, eV, edV, g_ok.sum()), flush=True)
            bad += int(f2)
            if f2 and os.environ.get("FUZZ_DUMP"):
                idx = np.nonzero(apart)[0]
                print("     apart instances", idx.tolist(), "product status", st[apart].tolist(), "iters", it[apart].tolist(), "port status", ref.status[apart].tolist(),
                      "port ipm", ref.ipm_iter[apart].tolist(), flush=True)
                np.savez(os.path.join(os.environ["FUZZ_DUMP"], "fuzz_%d_%s.npz" % (seed, tag.replace(" ", "_").replace("=", ""))), idx=idx, **DUMP)
        keep = ~apart
        import dataclasses
        ref = cpu_port.PortResult(*[getattr(ref, f.name)[keep] if getattr(ref, f.name) is not None else None for f in dataclasses.fields(cpu_port.PortResult)])
        class _R: pass
        rr = _R()
        kt = torch.as_tensor(keep, device=r.status.device)
        for n in ("u0", "V", "status", "iters", "dV_dp", "dpi_dp"):
            v = getattr(r, n)
            setattr(rr, n, v[kt] if v is not None else None)
        r, st, it = rr, st[keep], it[keep]
    ok = (st == 0) & (ref.status == 0)
    if rti:       # one SQP iteration: status 2 is the regular outcome, the step taken is what is compared
        ok = (st == ref.status) & ((st == 0) | (st == 2))
    e = {"u0": rel(r.u0.cpu().numpy()[ok], ref.u0[ok]), "V": rel(r.V.cpu().numpy()[ok], ref.V[ok])}
    if r.dV_dp is not None:
        e["dV"] = rel(r.dV_dp.cpu().numpy()[ok], ref.dV[ok])
    if r.dpi_dp is not None:
        dk, dr = r.dpi_dp.cpu().numpy()[ok], ref.dpi[ok]
        fin = ~(np.isnan(dk).reshape(len(dk), -1).any(1) | np.isnan(dr).reshape(len(dr), -1).any(1))
        e["dpi"] = rel(dk[fin], dr[fin])
    worst = {k: (float(v.max()) if len(v) else 0.0) for k, v in e.items()}
    same = float((st == ref.status).mean())
    dsqp = int(np.abs(it[ok, 0] - ref.sqp_iter[ok]).max()) if ok.any() else 0
    fail = same < 0.995 or dsqp > 1 or any(v > (dpi_tol if k == "dpi" else 1e-6) for k, v in worst.items())
    print("%-4s %s: status equal %.4f, conv %.3f, |d sqp| <= %d, %s" % ("FAIL" if fail else "ok", tag, same, float((st == 0).mean()), dsqp,
          " ".join("%s %.1e" % kv for kv in worst.items())), flush=True)
    bad += int(fail)
    if fail and os.environ.get("FUZZ_DUMP") and among is None:
        np.savez(os.path.join(os.environ["FUZZ_DUMP"], "fuzz_%d_%s.npz" % (seed, tag.replace(" ", "_").replace("=", ""))), ok=ok, st=st, it=it,
                 u0=r.u0.cpu().numpy(), V=r.V.cpu().numpy(), dV=r.dV_dp.cpu().numpy() if r.dV_dp is not None else 0, **DUMP)
        for k, v in e.items():
            if len(v):
                j = int(np.argmax(v))
                gi = np.nonzero(ok)[0][j] if k != "dpi" else -1
                print("     worst %s %.2e at converged instance %d: iters product %s port (%d, %d)" % (k, v[j], gi, it[gi].tolist(), ref.sqp_iter[gi], ref.ipm_iter[gi]), flush=True)
    return both


for trial in range(int(os.environ.get("FUZZ_TRIALS", "24"))):
    fam = ("cartpole", "linear", "chain")[trial % 3]
    if fam == "cartpole":
        N = int(rng.choice([3, 7, 12, 15, 20, 21, 31, 40, 63]))
        B = int(rng.choice([1, 2, 5, 63, 64, 65, 257, 1000]))
        ocp, P = cartpole_ocp(N=N, tf=0.1 * N), make_cartpole(N=N, tf=0.1 * N)
        x0 = rng.uniform(-1, 1, (B, 4)) * np.array([0.8, 1.5, 0.5, 1.5])
        x0[: B // 3, :] = 0.0
        x0[: B // 3, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B // 3)
        dpi_tol = 1e-5
    elif fam == "linear":
        N = int(rng.choice([2, 5, 13, 23, 24, 40, 47, 48, 55, 63]))
        B = int(rng.choice([1, 3, 8, 64, 129, 1000]))
        g = float(rng.choice([1.0, 0.99, 0.9]))
        ocp, P = linear_system_ocp(discount_factor=g, N=N), make_linear_system(gamma=g, N=N)
        x0 = np.column_stack([rng.uniform(0.1, 0.9, B), rng.uniform(-0.6, 0.6, B)])
        dpi_tol = 1e-3            # (weakly active soft rows: see tests/test_gpu_fullsize.py)
    else:
        nm = int(rng.choice([3, 4, 5, 6, 7]))
        N = int(rng.choice([5, 17, 40, 50]))
        B = int(rng.choice([1, 3, 17, 64])) if nm < 7 else int(rng.choice([1, 3, 5]))
        ocp, P = chain_mass_ocp(n_mass=nm, N=N), make_chain_mass(n_mass=nm, N=N)
        M = nm - 2
        x0 = np.tile(ocp.x0, (B, 1))
        x0[:, 3 * (M + 1):] += rng.normal(0.0, 1e-2, (B, 3 * M))
        dpi_tol = 1e-5
    per = int(rng.integers(3))      # 0: shared parameters; 1: the model block per instance; 2: every parameter per instance (cost blocks too)
    theta = np.tile(P.p0, (B, 1))
    nm_p = ocp.n_model_p if hasattr(ocp, "n_model_p") else 3
    if per == 1:
        theta[:, :nm_p] *= rng.uniform(0.97, 1.03, (B, nm_p))
    elif per == 2:
        theta *= rng.uniform(0.97, 1.03, theta.shape)
        if fam == "linear":
            theta[:, 6:8] += rng.normal(0.0, 0.02, (B, 2))      # b
            theta[:, 9:] += rng.normal(0.0, 0.05, (B, 3))       # f
    mpc = MPCBatch(ocp, B)
    if per:
        mpc.set_theta(torch.as_tensor(theta))
    tag = "%s N=%d B=%d per-instance-theta=%d" % (fam if fam != "chain" else "chain%d" % nm, N, B, per)
    r = mpc.solve(x0, sens_v=True, sens_pi=True, cold=True)
    ref = cpu_port.solve(P, x0, p=theta if per else None)
    uq = (lambda xx, **kw: cpu_port.solve(P, xx, p=theta if per else None, exact=True, **kw)) if fam == "linear" else None
    check(tag + " cold", r, ref, dpi_tol, unique=uq(x0) if uq else None)
    # a warm call at moved states with a per-instance cold mask
    x1 = x0 + rng.normal(0, 0.01, x0.shape)
    if fam == "linear":
        x1 = np.clip(x1, [0.05, -0.9], [0.95, 0.9])
    mask = rng.uniform(size=B) < 0.3
    DUMP = dict(x0=x0, x1=x1, mask=mask, theta=theta, N=N, B=B, per=per, g=g if fam == "linear" else 0.0)
    r1 = mpc.solve(x1, sens_v=True, sens_pi=True, cold_mask=torch.as_tensor(mask, device="cuda"))
    ref_c = cpu_port.solve(P, x1, p=theta if per else None)
    ref_w = cpu_port.solve(P, x1, p=theta if per else None, warm=ref)
    mix = cpu_port.PortResult(*[np.where(mask.reshape((-1,) + (1,) * (getattr(ref_c, f.name).ndim - 1)), getattr(ref_c, f.name), getattr(ref_w, f.name))
                                 if getattr(ref_c, f.name) is not None else None for f in __import__("dataclasses").fields(cpu_port.PortResult)])
    check(tag + " warm + cold mask", r1, mix, dpi_tol, unique=uq(x1) if uq else None)
    # a closed loop from there: two more warm calls at larger moves, then one real-time iteration (FUZZ_CHAIN=0: skip)
    if os.environ.get("FUZZ_CHAIN", "1") != "0":
        xs, prev = x1, mix
        solved = (r1.status.cpu().numpy() == 0) & (mix.status == 0)
        for n, sc in enumerate((3.0, 1.0)):
            xs = xs + rng.normal(0, 0.01 * sc * (0.3 if fam == "chain" else 1.0), xs.shape)
            if fam == "linear":
                xs = np.clip(xs, [0.05, -0.9], [0.95, 0.9])
            DUMP = dict(x0=x0, x1=xs, mask=mask, theta=theta, N=N, B=B, per=per)
            rn = mpc.solve(xs, sens_v=True, sens_pi=True)
            prev = cpu_port.solve(P, xs, p=theta if per else None, warm=prev)
            solved &= check(tag + " warm %d" % (n + 2), rn, prev, dpi_tol, unique=uq(xs) if uq else None, among=solved.copy())
        if fam != "linear":      # (an RTI step of the LQ model is its solve)
            xs = xs + rng.normal(0, 0.003, xs.shape)
            rn = mpc.solve(xs, sens_pi=True, rti=True)
            prev = cpu_port.solve(P, xs, p=theta if per else None, warm=prev, rti=True)
            check(tag + " rti", rn, prev, dpi_tol, rti=True, among=solved.copy())
    # Q-mode
    lo, hi = np.asarray(ocp.lbu, float), np.asarray(ocp.ubu, float)
    u0 = rng.uniform(0.8 * lo, 0.8 * hi, (B, len(lo)))
    rq = mpc.solve(x1, u0, sens_v=True, cold=True)
    refq = cpu_port.solve(P, x1, p=theta if per else None, u0fix=u0, flags=1)
    check(tag + " Q-mode", rq, refq, dpi_tol, unique=uq(x1, u0fix=u0, flags=1) if uq else None)
print("fuzz seed", seed, "mismatching configurations:", bad)
sys.exit(1 if bad else 0)
