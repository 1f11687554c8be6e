"""G7c: third-party GRADIENTS of the cartpole NLP at 96 states — breadth.  `python tests/golden/make_thirdparty_grad3.py` (~10 minutes, 3 cores).

G7 / G7b hold central differences of scipy-SLSQP solutions at 4 + 10 cartpole states (SLSQP needs minutes per state and parameter).  Here the
KKT points come from MINPACK's Powell hybrid method (scipy.optimize.root, "hybr", analytic Jacobian) on the KKT EQUATIONS of the NLP with
the active set of the solution — the method of G8's n_mass 5 / 7 rows (make_thirdparty_chain_grad.kkt_root) — which takes seconds, so the
set can be wide:
  * 32 swing-up starts of the reference's reset distribution (theta ~ U(0.9 pi, 1.1 pi), continuous_cartpole/environment.py:178-180):
    u is on its bound over the first stages (du0*/dp = 0 exactly; dV/dp is not),
  * 32 states from the box +-[0.5, 1, 0.3, 1] (near upright, u0* inside its bounds),
  * 32 states from the wider box +-[1.8, 2.5, 0.6, 2] (state bounds active on part of them),
each at the nominal parameters and over (M, m, l) at delta = 1e-5 and 1e-4.  The starting point of the root solve at p is the KKT point the
C++ port finds; the points at p (1 +- delta) start from the root at p.  Every point is certified by make_thirdparty.certify (multipliers by
least squares, stationarity, feasibility, signs); a state is KEPT when its base point is strictly complementary (smallest active
multiplier > 1e-4, every inactive bound further than 1e-6 away) and the perturbed points have the same active set — the others
are dropped and counted (kept / drawn is stored).  Inputs and expected outputs only (g7c_cartpole_grad.npz)."""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_thirdparty import Nlp, certify  # noqa: E402
from make_thirdparty_chain_grad import kkt_root  # noqa: E402
from oracle.problems import make_cartpole  # noqa: E402

torch.set_num_threads(1)
PARTS = "/tmp/g7c_parts"
DELTA = (1e-5, 1e-4)


def states():
    rng = np.random.default_rng(2026)
    a = np.zeros((32, 4))
    a[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, 32)
    b = rng.uniform(-1, 1, (32, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    c = rng.uniform(-1, 1, (32, 4)) * np.array([1.8, 2.5, 0.6, 2.0])
    return np.vstack([a, b, c])


def job(i):
    from oracle import cpu_port
    f = os.path.join(PARTS, f"{i}.pkl")
    if os.path.exists(f):
        return i
    P = make_cartpole()
    x0, p0 = states()[i], P.p0.copy()
    out = {"keep": False}
    r = cpu_port.solve(P, x0[None], tol=1e-9, flags=0)
    if r.status[0] == 0:
        nlp = Nlp(P, x0, p0)
        z0 = np.concatenate([r.U[0].ravel(), r.X[0, 1:].ravel()])
        zb, vb, wb, resb, _ = kkt_root(nlp, z0)
        kb = certify(nlp, zb)
        margin = np.minimum(zb - nlp.lo, nlp.hi - zb)
        inactive_margin = float(margin[margin > 1e-7].min()) if (margin > 1e-7).any() else 1.0
        strict = (kb["n_active"] == 0 or kb["min_multiplier"] > 1e-4) and inactive_margin > 1e-6 and resb < 1e-10 and kb["stationarity"] < 1e-9
        out.update(u0=zb[: P.nu].copy(), V=vb, kkt=[kb["stationarity"], kb["feasibility"], kb["min_multiplier"]], n_active=kb["n_active"],
                   s_max=float(np.abs(zb[P.N * P.nu:].reshape(P.N, P.nx)[:, 0]).max()))
        ok = strict
        for d in DELTA:
            dV, du, st = np.zeros(3), np.zeros((3, P.nu)), 0.0
            for j in range(3):
                pp, pm = p0.copy(), p0.copy()
                pp[j] *= 1.0 + d
                pm[j] *= 1.0 - d
                zp, vp, _, rp, _ = kkt_root(Nlp(P, x0, pp), zb, wb)
                zm, vm, _, rm, _ = kkt_root(Nlp(P, x0, pm), zb, wb)
                kp, km = certify(Nlp(P, x0, pp), zp), certify(Nlp(P, x0, pm), zm)
                ok = ok and kp["n_active"] == kb["n_active"] == km["n_active"] and max(rp, rm) < 1e-10 and max(kp["stationarity"], km["stationarity"]) < 1e-9 \
                    and (kb["n_active"] == 0 or min(kp["min_multiplier"], km["min_multiplier"]) > 0.0) and max(kp["feasibility"], km["feasibility"]) < 1e-9
                dV[j], du[j], st = (vp - vm) / (2 * d * p0[j]), (zp[: P.nu] - zm[: P.nu]) / (2 * d * p0[j]), max(st, kp["stationarity"], km["stationarity"])
            out[d] = (dV, du, st)
        out["keep"] = bool(ok)
    print("state", i, x0.round(3), "keep", out["keep"], "u0", out.get("u0"), "n_active", out.get("n_active"), flush=True)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(out, fh)
    os.replace(f + ".tmp", f)
    return i


def main(assemble_only=False, procs=int(os.environ.get("G7C_PROCS", "3"))):
    import multiprocessing as mp
    os.makedirs(PARTS, exist_ok=True)
    X = states()
    if not assemble_only:
        with mp.get_context("spawn").Pool(procs) as pool:
            for _ in pool.imap_unordered(job, range(len(X)), chunksize=1):
                pass
    res = [pickle.load(open(os.path.join(PARTS, f"{i}.pkl"), "rb")) for i in range(len(X))]
    keep = np.array([r["keep"] for r in res])
    idx = np.where(keep)[0]
    out = {"delta": np.array(DELTA), "drawn": np.array(len(X)), "kept_index": idx, "x0": X[idx],
           "u0": np.array([res[i]["u0"] for i in idx]), "V": np.array([res[i]["V"] for i in idx]), "kkt": np.array([res[i]["kkt"] for i in idx]),
           "n_active": np.array([res[i]["n_active"] for i in idx]), "s_max": np.array([res[i]["s_max"] for i in idx])}
    for di, d in enumerate(DELTA):
        out[f"dV_d{di}"] = np.array([res[i][d][0] for i in idx])
        out[f"du0_d{di}"] = np.array([res[i][d][1] for i in idx])          # [state, param, nu]
        out[f"kkt_d{di}"] = np.array([res[i][d][2] for i in idx])
    np.savez(os.path.join(HERE, "g7c_cartpole_grad.npz"), **out)
    print("kept", len(idx), "of", len(X), "| u0 saturated:", int((np.abs(out["u0"]) > 29.999).sum()), "| state bound active:", int((np.abs(out["s_max"] - 2.4) < 1e-7).sum()))


if __name__ == "__main__":
    main(assemble_only=len(sys.argv) > 1 and sys.argv[1] == "assemble")
