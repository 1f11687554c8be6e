"""G7b: third-party GRADIENTS of the cartpole NLP at more states, some with an ACTIVE STATE BOUND on the horizon.
`python tests/golden/make_thirdparty_grad2.py` (dev container, ~20 minutes on 5 cores; resumable, parts under /tmp/g7b_parts).

G7 (make_thirdparty_grad.py) holds central differences of scipy-SLSQP's V and u0* over (M, m, l) at 4 cartpole states, two of them
swing-up starts where u0* sits on its bound and du0*/dp is identically zero, none with a state bound active.  Here: 10 states where
u0* is strictly inside [-30, 30]:
  * 4 with the cart close to the end of the track and moving towards it (the bound |s| <= 2.4 of config/cartpole.yaml:82-91 is ACTIVE on
    stages of the horizon — the generator asserts it on SLSQP's own solution: some |s_k| within 1e-7 of 2.4),
  * 6 near-upright states drawn once (seed 77) from the box +-[0.5, 1, 0.3, 1].
Same recipe as G7: SLSQP (code this repository did not write) from the reference's cold iterate at p, then at p (1 +- delta) from the
solution at p, delta = 1e-5 and 1e-4, every point KKT-certified by make_thirdparty.certify.  Inputs and expected outputs only
(g7b_cartpole_grad.npz)."""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_thirdparty import Nlp, certify  # noqa: E402
from make_thirdparty_grad import cold, slsqp  # noqa: E402
from oracle.problems import make_cartpole  # noqa: E402

torch.set_num_threads(1)
# G7B_METHOD=kkt: the same ten states through MINPACK's hybrid method on the KKT equations (make_thirdparty_chain_grad.kkt_root: seconds
# per point where SLSQP needs 10-25 minutes per state and parameter) — the fixture records which method produced it ("method")
METHOD = os.environ.get("G7B_METHOD", "slsqp")
PARTS = "/tmp/g7b_parts" if METHOD == "slsqp" else "/tmp/g7b_parts_kkt"
DELTA = (1e-5, 1e-4)
# cart near the end of the track, moving outwards (found with the CPU port: position bound active on 1-4 stages, |u0*| <= 12)
ACTIVE = np.array([[2.288, 1.838, -0.174, -0.525], [-1.952, -2.336, -0.241, 0.521], [2.231, 1.008, 0.25, 0.475], [-2.155, -2.153, 0.032, -0.182]])


def states():
    rng = np.random.default_rng(77)
    free = rng.uniform(-1, 1, (6, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    return np.vstack([ACTIVE, free])


def job(item):
    i, j = item
    f = os.path.join(PARTS, f"{i}_{j}.pkl")
    if os.path.exists(f):
        return item
    P = make_cartpole()
    x0, p0 = states()[i], P.p0.copy()
    nlp = Nlp(P, x0, p0)
    wb = None
    if METHOD == "kkt":
        from make_thirdparty_chain_grad import kkt_root
        from oracle import cpu_port
        r = cpu_port.solve(P, x0[None], tol=1e-9, flags=0)
        assert r.status[0] == 0
        zb, vb, wb, resb, _ = kkt_root(nlp, np.concatenate([r.U[0].ravel(), r.X[0, 1:].ravel()]))
        assert resb < 1e-10
    else:
        zb, vb, _ = slsqp(nlp, cold(nlp, x0))
    kb = certify(nlp, zb)
    xs = zb[P.N * P.nu: P.N * P.nu + P.N * P.nx].reshape(P.N, P.nx)
    out = {"u0": zb[: P.nu].copy(), "V": vb, "kkt": [kb["stationarity"], kb["feasibility"], kb["min_multiplier"]], "n_active": kb["n_active"],
           "s_max": float(np.abs(xs[:, 0]).max()), "u_max": float(np.abs(zb[: P.N * P.nu]).max())}
    for d in DELTA:
        pp, pm = p0.copy(), p0.copy()
        pp[j] *= 1.0 + d
        pm[j] *= 1.0 - d
        if METHOD == "kkt":
            zp, vp, _, rp, _ = kkt_root(Nlp(P, x0, pp), zb, wb)
            zm, vm, _, rm, _ = kkt_root(Nlp(P, x0, pm), zb, wb)
            assert max(rp, rm) < 1e-10
        else:
            zp, vp, _ = slsqp(Nlp(P, x0, pp), zb)
            zm, vm, _ = slsqp(Nlp(P, x0, pm), zb)
        kp, km = certify(Nlp(P, x0, pp), zp), certify(Nlp(P, x0, pm), zm)
        out[d] = ((vp - vm) / (2 * d * p0[j]), (zp[: P.nu] - zm[: P.nu]) / (2 * d * p0[j]), max(kp["stationarity"], km["stationarity"]))
    print("cartpole state", i, "param", j, "u0", out["u0"], "max |s|", out["s_max"], {d: out[d][:2] for d in DELTA}, out["kkt"], flush=True)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(out, fh)
    os.replace(f + ".tmp", f)
    return item


def main(assemble_only=False, procs=int(os.environ.get("G7B_PROCS", "5"))):
    import multiprocessing as mp
    os.makedirs(PARTS, exist_ok=True)
    X = states()
    items = [(i, j) for i in range(len(X)) for j in range(3)]
    if not assemble_only:
        with mp.get_context("spawn").Pool(procs) as pool:
            for it in pool.imap_unordered(job, items, chunksize=1):
                print("done", it, flush=True)
    res = {it: pickle.load(open(os.path.join(PARTS, f"{it[0]}_{it[1]}.pkl"), "rb")) for it in items}
    n = len(X)
    out = {"method": np.array(METHOD), "delta": np.array(DELTA), "x0": X, "u0": np.array([res[(i, 0)]["u0"] for i in range(n)]), "V": np.array([res[(i, 0)]["V"] for i in range(n)]),
           "kkt": np.array([res[(i, 0)]["kkt"] for i in range(n)]), "s_max": np.array([res[(i, 0)]["s_max"] for i in range(n)]),
           "u_max": np.array([res[(i, 0)]["u_max"] for i in range(n)]), "n_active": np.array([res[(i, 0)]["n_active"] for i in range(n)])}
    for di, d in enumerate(DELTA):
        out[f"dV_d{di}"] = np.array([[res[(i, j)][d][0] for j in range(3)] for i in range(n)])
        out[f"du0_d{di}"] = np.array([[res[(i, j)][d][1] for j in range(3)] for i in range(n)])      # [state, param, nu]
        out[f"kkt_d{di}"] = np.array([[res[(i, j)][d][2] for j in range(3)] for i in range(n)])
    # what the fixture is for: u0* strictly inside its bounds everywhere, a state bound active on the first four
    assert np.all(np.abs(out["u0"]) < 29.0), out["u0"]
    assert np.all(np.abs(out["s_max"][: len(ACTIVE)] - 2.4) < 1e-7), out["s_max"]
    assert out["kkt"][:, :2].max() < 1e-9 and out["kkt_d0"].max() < 1e-9 and np.all(out["n_active"][: len(ACTIVE)] >= 1)
    np.savez(os.path.join(HERE, "g7b_cartpole_grad.npz"), **out)
    print("wrote g7b_cartpole_grad.npz: max |s| on the horizon", out["s_max"], "u0", out["u0"].ravel())


if __name__ == "__main__":
    main(assemble_only=len(sys.argv) > 1 and sys.argv[1] == "assemble")
