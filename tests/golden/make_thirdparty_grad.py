"""G7: third-party GRADIENTS and third-party CHAIN solutions.  `python tests/golden/make_thirdparty_grad.py` (dev container: ~15 minutes on 8 cores with one BLAS thread per worker — OPENBLAS_NUM_THREADS=1 — plus 4.4 h for the chain n_mass 5 solve; resumable).

G6 (make_thirdparty.py) holds u0*, V of KKT points found by scipy.optimize for cartpole and the linear system.  Here the same
third-party solves are differentiated by CENTRAL DIFFERENCES over the parameters — the reference's own sanity check of dpi/dp is
finite differences along a parameter sweep (rlmpc/examples/chain_mass.py:28-64, scripts/linear_system_mpc_nlp.py:17-106) — so that
dV/dp and du0*/dp of the product are held to numbers no code of this repository produced:

  * cartpole: 4 states of G6 (2 swing-up starts, 2 near upright) x the 3 model parameters (M, m, l): SLSQP at p (1 +- delta), started from the
    solution at p, delta = 1e-5 (and 1e-4 as a noise check: both estimates are stored, the tests use 1e-5 where the two agree to 2e-6);
  * linear system: 2 states with an unsaturated u0* x 12 parameters.  The QP at p is solved by trust-constr and then POLISHED: with the
    active set trust-constr ends on, the KKT conditions of a QP are one linear system (numpy); the QPs at p +- delta start the polish from
    the base point's active set.  Every point is certified (stationarity, feasibility <= 1e-12, multiplier signs) before it is used —
    a strictly convex QP has one KKT point, so the certificate, not the route, makes it the solution.  The polish also brings G6's linear
    rows from the 2e-6 stationarity trust-constr stops at to 1e-13 (stored as lin_*_polished);
  * chain of masses n_mass 3 and 5 (N = 40: 480 / 960 unknowns, bounds on the controls only): SLSQP from the reference's cold iterate,
    KKT-certified; u0*, V.

Inputs and expected outputs only (g7_thirdparty_grad.npz).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_thirdparty import Nlp, certify  # noqa: E402
from oracle.problems import make_cartpole, make_chain_mass, make_linear_system  # noqa: E402

torch.set_num_threads(1)


def slsqp(nlp, z0, max_rounds=4):
    from scipy.optimize import minimize
    t = lambda z: torch.tensor(z, dtype=torch.float64)
    cons = [{"type": "eq", "fun": lambda z: nlp.g(t(z)).numpy(), "jac": lambda z: nlp.Jg(t(z)).numpy()}]
    bounds = list(zip(nlp.lo, nlp.hi))
    z, its = np.clip(z0, nlp.lo, nlp.hi), 0
    for _ in range(max_rounds):      # SLSQP stops on the change of f: restart it from its answer until it stays
        r = minimize(lambda z_: float(nlp.f(t(z_))), z, jac=lambda z_: nlp.gradf(t(z_)).numpy(), bounds=bounds, constraints=cons, method="SLSQP",
                     options={"ftol": 1e-16, "maxiter": 3000})
        its += r.nit
        done = np.abs(r.x - z).max() < 1e-12
        z = r.x
        if done:
            break
    return z, float(nlp.f(t(z))), its


def cold(nlp, x0):
    P = nlp.P
    z0 = np.zeros(nlp.nz)
    z0[P.N * P.nu: P.N * P.nu + P.N * P.nx] = np.tile(x0, P.N)      # MPC.reset: x_k = x0, u = 0
    return z0


def cartpole_job(job):
    x0, p0, j, delta = job
    P = make_cartpole()
    nlp = Nlp(P, x0, p0)
    zb, vb, _ = slsqp(nlp, cold(nlp, x0))
    kkt = certify(nlp, zb)
    out = {"u0": zb[: P.nu].copy(), "V": vb, "kkt": [kkt["stationarity"], kkt["feasibility"], kkt["min_multiplier"]]}
    for d in delta:
        pp, pm = p0.copy(), p0.copy()
        pp[j] *= 1.0 + d
        pm[j] *= 1.0 - d
        zp, vp, _ = slsqp(Nlp(P, x0, pp), zb)
        zm, vm, _ = slsqp(Nlp(P, x0, pm), zb)
        kp, km = certify(Nlp(P, x0, pp), zp), certify(Nlp(P, x0, pm), zm)
        out[d] = ((vp - vm) / (2 * d * p0[j]), (zp[: P.nu] - zm[: P.nu]) / (2 * d * p0[j]), max(kp["stationarity"], km["stationarity"]))
    print("cartpole", x0, "param", j, {d: (out[d][0], out[d][1]) for d in delta}, flush=True)
    return out


def qp_polish(nlp, z, act_tol=1e-6):
    """KKT point of the QP on the active set z ends on: [H A'; A 0] [z; mult] = [-c; b] with A = equalities + active bounds + active
    soft rows, exact to rounding; the active set is repaired until feasibility and the multiplier signs hold."""
    t = torch.tensor(z)
    H = torch.autograd.functional.hessian(nlp.f, t).numpy()
    c = nlp.gradf(t).numpy() - H @ z
    Jg = nlp.Jg(t).numpy()
    g0 = nlp.g(t).numpy() - Jg @ z
    Jh = torch.autograd.functional.jacobian(nlp.h, t).numpy() if nlp.ns else np.zeros((0, nlp.nz))
    h0 = nlp.h(t).numpy() - Jh @ z if nlp.ns else np.zeros(0)
    act_lo = set(np.where(z - nlp.lo < act_tol)[0])
    act_hi = set(np.where(nlp.hi - z < act_tol)[0])
    act_h = set(np.where(nlp.h(t).numpy() < act_tol)[0]) if nlp.ns else set()
    for _ in range(50):
        rows, rhs, kind = [Jg], [-g0], []
        for i in sorted(act_lo):
            e = np.zeros(nlp.nz); e[i] = 1.0; rows.append(e[None]); rhs.append([nlp.lo[i]]); kind.append(("lo", i))
        for i in sorted(act_hi):
            e = np.zeros(nlp.nz); e[i] = 1.0; rows.append(e[None]); rhs.append([nlp.hi[i]]); kind.append(("hi", i))
        for i in sorted(act_h):
            rows.append(Jh[i][None]); rhs.append([-h0[i]]); kind.append(("h", i))
        A = np.vstack(rows)
        b = np.concatenate(rhs)
        n, m = nlp.nz, A.shape[0]
        K = np.block([[H, A.T], [A, np.zeros((m, m))]])
        sol = np.linalg.lstsq(K, np.concatenate([-c, b]), rcond=None)[0]
        zn, mult = sol[:n], sol[n + Jg.shape[0]:]
        # multiplier signs: H z + c + A' mult = 0; a lower bound / soft row (>= side) wants mult <= 0, an upper bound mult >= 0
        changed = False
        for (kd, i), mu in zip(kind, mult):
            bad = mu > 1e-12 if kd in ("lo", "h") else mu < -1e-12
            if bad:
                {"lo": act_lo, "hi": act_hi, "h": act_h}[kd].discard(i)
                changed = True
        viol_lo, viol_hi = np.where(nlp.lo - zn > 1e-12)[0], np.where(zn - nlp.hi > 1e-12)[0]
        hv = Jh @ zn + h0 if nlp.ns else np.zeros(0)
        for i in viol_lo:
            act_lo.add(i); changed = True
        for i in viol_hi:
            act_hi.add(i); changed = True
        for i in np.where(hv < -1e-12)[0]:
            act_h.add(i); changed = True
        z = zn
        if not changed:
            break
    return z


def linear_solve(P, x0, p, gamma, z_start=None):
    from scipy.optimize import Bounds, LinearConstraint, minimize
    nlp = Nlp(P, x0, p, gamma)
    t = lambda z: torch.tensor(z, dtype=torch.float64)
    if z_start is not None:
        # a perturbed problem: the active set of the base solution is the starting guess of the polish (a strictly convex QP has ONE KKT
        # point: whatever finds it, the certificate below is what makes it the solution)
        z = qp_polish(nlp, z_start)
        return z, float(nlp.f(t(z))), certify(nlp, z, act_tol=1e-9), nlp
    z0 = np.clip(cold(nlp, x0) if z_start is None else z_start, nlp.lo, nlp.hi)
    H = torch.autograd.functional.hessian(nlp.f, t(z0)).numpy()
    Jg = nlp.Jg(t(z0)).numpy()
    g0 = nlp.g(t(z0)).numpy() - Jg @ z0
    Jh = torch.autograd.functional.jacobian(nlp.h, t(z0)).numpy()
    h0 = nlp.h(t(z0)).numpy() - Jh @ z0
    r = minimize(lambda z: float(nlp.f(t(z))), z0, jac=lambda z: nlp.gradf(t(z)).numpy(), hess=lambda z: H, method="trust-constr",
                 bounds=Bounds(nlp.lo, nlp.hi), constraints=[LinearConstraint(Jg, -g0, -g0), LinearConstraint(Jh, -h0, np.inf)],
                 options=dict(gtol=1e-13, xtol=1e-15, barrier_tol=1e-13, maxiter=3000, initial_barrier_parameter=0.1, initial_barrier_tolerance=0.1))
    z = qp_polish(nlp, r.x)
    kkt = certify(nlp, z, act_tol=1e-9)
    return z, float(nlp.f(t(z))), kkt, nlp


def linear_job(job):
    x0, gamma, j, delta = job
    P = make_linear_system(gamma=gamma)
    p0 = P.p0.copy()
    zb, vb, kkt, _ = linear_solve(P, x0, p0, gamma)
    out = {"u0": zb[: P.nu].copy(), "V": vb, "kkt": [kkt["stationarity"], kkt["feasibility"], kkt["min_multiplier"]]}
    if j is None:
        print("linear polished", gamma, x0, out, flush=True)
        return out
    scale = max(abs(p0[j]), 1.0)         # (b, V_0, f are zero or tiny at the nominal point: absolute steps there)
    for d in delta:
        pp, pm = p0.copy(), p0.copy()
        pp[j] += d * scale
        pm[j] -= d * scale
        zp, vp, kp, _ = linear_solve(P, x0, pp, gamma, zb)
        zm, vm, km, _ = linear_solve(P, x0, pm, gamma, zb)
        out[d] = ((vp - vm) / (2 * d * scale), (zp[: P.nu] - zm[: P.nu]) / (2 * d * scale), max(kp["stationarity"], km["stationarity"]))
    print("linear", gamma, x0, "param", j, {d: (out[d][0], out[d][1]) for d in delta}, "kkt", out["kkt"], flush=True)
    return out


def chain_job(job):
    n_mass, x0 = job
    P = make_chain_mass(n_mass=n_mass)
    nlp = Nlp(P, x0, P.p0)
    z, v, its = slsqp(nlp, cold(nlp, x0), max_rounds=6)
    kkt = certify(nlp, z)
    print("chain", n_mass, "u0", z[: P.nu], "V", v, "its", its, kkt, flush=True)
    return z[: P.nu].copy(), v, [kkt["stationarity"], kkt["feasibility"], kkt["min_multiplier"]]


PARTS = "/tmp/g7_parts"


def _run(item):
    """One job; its result is pickled under PARTS as soon as it is there (the chain n_mass 5 solve runs for hours: a later
    `make_thirdparty_grad.py assemble` builds the file from whatever is complete)."""
    import pickle
    idx, (kind, payload) = item
    f = os.path.join(PARTS, f"{idx}.pkl")
    if os.path.exists(f):
        return idx
    if kind == "chain" and payload[0] == 5 and os.environ.get("G7_SKIP_CHAIN5"):      # (SLSQP on its 960 unknowns did not finish in 3 h)
        return idx
    r = {"cartpole": cartpole_job, "linear": linear_job, "chain": chain_job}[kind](payload)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(r, fh)
    os.replace(f + ".tmp", f)
    return idx


def job_list():
    delta = (1e-5, 1e-4)
    g6 = np.load(os.path.join(HERE, "g6_thirdparty.npz"))
    Pc = make_cartpole()
    cp_rows = [8, 11, 9, 10]                  # near-upright states of G6 (nominal parameters): u0* is not saturated there
    jobs = []
    for r in cp_rows:
        p = Pc.p0.copy()
        p[:3] = g6["cp_theta_model"][r]
        jobs += [("cartpole", (g6["cp_x0"][r], p, j, delta)) for j in range(3)]
    lin_x0 = np.array([[0.2, 0.2], [0.15, 0.1]])         # u0* interior (at [0.5, 0.5] the control saturates: du0*/dp = 0), moving away from the soft bound x[0] >= 0
    for gamma in (0.99, 0.9):
        jobs += [("linear", (x0, gamma, j, delta)) for x0 in lin_x0 for j in range(12)]
    for gamma, tag in ((0.99, "g099"), (0.9, "g09")):      # G6's linear rows, polished
        jobs += [("linear", (x0, gamma, None, delta)) for x0 in g6[f"lin_{tag}_x0"]]
    chain_x0 = {}
    rng = np.random.default_rng(31)
    from mpc4rl_amd.problems import chain_mass_ocp
    for n_mass in (3, 5):
        ocp = chain_mass_ocp(n_mass=n_mass)
        M = n_mass - 2
        x0 = ocp.x0.copy()
        x0[3 * (M + 1):] += rng.normal(0.0, 1e-2, 3 * M)
        chain_x0[n_mass] = x0
        jobs.append(("chain", (n_mass, x0)))
    return jobs, delta, g6, cp_rows, lin_x0, chain_x0


def main(procs=int(os.environ.get("G7_PROCS", "7")), assemble_only=False):
    import multiprocessing as mp
    import pickle
    os.makedirs(PARTS, exist_ok=True)
    jobs, delta, g6, cp_rows, lin_x0, chain_x0 = job_list()
    if not assemble_only:
        # the long jobs first
        order = sorted(range(len(jobs)), key=lambda i: {"chain": 0, "cartpole": 1, "linear": 2}[jobs[i][0]])
        with mp.get_context("spawn").Pool(procs) as pool:
            for idx in pool.imap_unordered(_run, [(i, jobs[i]) for i in order], chunksize=1):
                print("done", idx, jobs[idx][0], flush=True)
    res = []
    for i in range(len(jobs)):
        f = os.path.join(PARTS, f"{i}.pkl")
        res.append(pickle.load(open(f, "rb")) if os.path.exists(f) else None)
    missing = [i for i, r in enumerate(res) if r is None]
    print("missing jobs:", [(i, jobs[i][0]) for i in missing])
    # a chain solve may be left out entirely; a gradient job of the linear system that did not finish leaves NaN entries (the tests use
    # the finite ones) — the generator takes hours on 8 cores and is resumable: finished jobs are kept in /tmp/g7_parts
    assert all(jobs[i][0] in ("chain", "linear") for i in missing), "cartpole jobs must all be there"
    out = {"delta": np.array(delta)}
    o = 0
    cp = res[o:o + 3 * len(cp_rows)]
    o += 3 * len(cp_rows)
    out["cp_x0"] = g6["cp_x0"][cp_rows]
    out["cp_theta_model"] = g6["cp_theta_model"][cp_rows]
    out["cp_u0"] = np.array([cp[3 * i]["u0"] for i in range(len(cp_rows))])
    out["cp_V"] = np.array([cp[3 * i]["V"] for i in range(len(cp_rows))])
    for di, d in enumerate(delta):
        out[f"cp_dV_d{di}"] = np.array([[cp[3 * i + j][d][0] for j in range(3)] for i in range(len(cp_rows))])
        out[f"cp_du0_d{di}"] = np.array([[cp[3 * i + j][d][1] for j in range(3)] for i in range(len(cp_rows))])      # [row, param, nu]
        out[f"cp_kkt_d{di}"] = np.array([[cp[3 * i + j][d][2] for j in range(3)] for i in range(len(cp_rows))])
    for gamma, tag in ((0.99, "g099"), (0.9, "g09")):
        part = res[o:o + 24]
        o += 24
        out[f"lin_{tag}_x0"] = lin_x0

        def first(i, key, shape):      # the nominal solve is repeated by every job of the state: any finished one has it
            for j in range(12):
                if part[12 * i + j] is not None:
                    return np.asarray(part[12 * i + j][key], float)
            return np.full(shape, np.nan)

        out[f"lin_{tag}_u0"] = np.array([first(i, "u0", (1,)) for i in range(2)])
        out[f"lin_{tag}_V"] = np.array([first(i, "V", ()) for i in range(2)])
        kk = [first(i, "kkt", (1,)) for i in range(2)]
        nk = max(len(np.atleast_1d(k)) for k in kk)
        out[f"lin_{tag}_kkt"] = np.array([np.resize(np.atleast_1d(k), nk) if np.all(np.isfinite(k)) else np.full(nk, np.nan) for k in kk])
        get = lambda i, j, d, k, shape: np.asarray(part[12 * i + j][d][k], float) if part[12 * i + j] is not None else np.full(shape, np.nan)
        for di, d in enumerate(delta):
            out[f"lin_{tag}_dV_d{di}"] = np.array([[get(i, j, d, 0, ()) for j in range(12)] for i in range(2)])
            out[f"lin_{tag}_du0_d{di}"] = np.array([[get(i, j, d, 1, (1,)) for j in range(12)] for i in range(2)])
            out[f"lin_{tag}_kkt_d{di}"] = np.array([[get(i, j, d, 2, ()) for j in range(12)] for i in range(2)])
    for gamma, tag in ((0.99, "g099"), (0.9, "g09")):
        n = len(g6[f"lin_{tag}_x0"])
        part = res[o:o + n]
        o += n
        have = [k for k, c in enumerate(part) if c is not None]
        out[f"lin_{tag}_polished_x0"] = g6[f"lin_{tag}_x0"][have]
        out[f"lin_{tag}_polished_u0"] = np.array([part[k]["u0"] for k in have]).reshape(len(have), 1)
        out[f"lin_{tag}_polished_V"] = np.array([part[k]["V"] for k in have]).reshape(len(have))
        out[f"lin_{tag}_polished_kkt"] = np.array([part[k]["kkt"] for k in have]).reshape(len(have), -1) if have else np.zeros((0, 1))
    for n_mass in (3, 5):
        if res[o] is not None:
            u0, v, kkt = res[o]
            out[f"chain{n_mass}_x0"], out[f"chain{n_mass}_u0"], out[f"chain{n_mass}_V"], out[f"chain{n_mass}_kkt"] = chain_x0[n_mass], u0, v, np.array(kkt)
        o += 1
    np.savez(os.path.join(HERE, "g7_thirdparty_grad.npz"), **out)
    print("wrote g7_thirdparty_grad.npz")


if __name__ == "__main__":
    main(assemble_only=len(sys.argv) > 1 and sys.argv[1] == "assemble")
