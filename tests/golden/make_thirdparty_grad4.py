"""G7d: third-party dQ/dp — the Q(s, a) mode of the path (MPC.q_update, rlmpc/mpc/common/mpc.py:52-96: lbu_0 = ubu_0 = u0, then dL/dp =
dQ/dp), cartpole.  `python tests/golden/make_thirdparty_grad4.py` (~5 minutes, 3 cores).

Q(s, a) is what the reference's Q-learning loop differentiates (examples/linear_system_mpc_qlearning.py:178-190), and until round 6 this
repository's dQ/dp was held to its own port and mirror only.  Here: 32 cartpole states (16 near upright, 16 from the box
+-[1.2, 2, pi, 3] that the port solves) with a pinned u0 ~ U(-25, 25); Q and its central differences over (M, m, l), delta = 1e-5 and 1e-4, of KKT
points found by MINPACK's hybrid method on the KKT equations with u_0 as an equality (make_thirdparty_chain_grad.kkt_root), each
certified (stationarity and feasibility by make_thirdparty.certify; the multiplier of the pinned u_0 has no sign).  Inputs and expected
outputs only (g7d_cartpole_qmode.npz)."""
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
PARTS = "/tmp/g7d_parts"
DELTA = (1e-5, 1e-4)


def cases():
    rng = np.random.default_rng(4242)
    a = rng.uniform(-1, 1, (16, 4)) * np.array([0.5, 1.0, 0.3, 1.0])
    b = rng.uniform(-1, 1, (16, 4)) * np.array([1.2, 2.0, np.pi, 3.0])
    return np.vstack([a, b]), rng.uniform(-25, 25, (32, 1))


def pinned(P, x0, p, u0):
    nlp = Nlp(P, x0, p)
    nlp.lo[: P.nu] = u0
    nlp.hi[: P.nu] = u0
    return nlp


def job(i):
    from oracle import cpu_port
    f = os.path.join(PARTS, f"{i}.pkl")
    if os.path.exists(f):
        return i
    P = make_cartpole()
    X, U0 = cases()
    x0, u0, p0 = X[i], U0[i], P.p0.copy()
    out = {"keep": False}
    r = cpu_port.solve(P, x0[None], u0fix=u0[None], tol=1e-9, flags=0)
    if r.status[0] == 0:
        nlp = pinned(P, x0, p0, u0)
        zb, vb, wb, resb, _ = kkt_root(nlp, np.concatenate([r.U[0].ravel(), r.X[0, 1:].ravel()]))
        kb = certify(nlp, zb)
        margin = np.minimum(zb - nlp.lo, nlp.hi - zb)[P.nu:]
        ok = resb < 1e-10 and kb["stationarity"] < 1e-9 and kb["feasibility"] < 1e-9 and (margin[margin > 1e-7].min() > 1e-6 if (margin > 1e-7).any() else True)
        out.update(Q=vb, kkt=[kb["stationarity"], kb["feasibility"]], n_active=kb["n_active"])
        for d in DELTA:
            dQ, st = np.zeros(3), 0.0
            for j in range(3):
                pp, pm = p0.copy(), p0.copy()
                pp[j] *= 1.0 + d
                pm[j] *= 1.0 - d
                zp, vp, _, rp, _ = kkt_root(pinned(P, x0, pp, u0), zb, wb)
                zm, vm, _, rm, _ = kkt_root(pinned(P, x0, pm, u0), zb, wb)
                kp, km = certify(pinned(P, x0, pp, u0), zp), certify(pinned(P, x0, pm, u0), zm)
                ok = ok and max(rp, rm) < 1e-10 and max(kp["stationarity"], km["stationarity"]) < 1e-9 and kp["n_active"] == kb["n_active"] == km["n_active"]
                dQ[j], st = (vp - vm) / (2 * d * p0[j]), max(st, kp["stationarity"], km["stationarity"])
            out[d] = (dQ, st)
        out["keep"] = bool(ok)
    print("case", i, x0.round(3), u0.round(2), "keep", out["keep"], out.get("Q"), flush=True)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(out, fh)
    os.replace(f + ".tmp", f)
    return i


def main(procs=int(os.environ.get("G7D_PROCS", "3"))):
    import multiprocessing as mp
    os.makedirs(PARTS, exist_ok=True)
    X, U0 = cases()
    with mp.get_context("spawn").Pool(procs) as pool:
        for _ in pool.imap_unordered(job, range(len(X)), chunksize=1):
            pass
    res = [pickle.load(open(os.path.join(PARTS, f"{i}.pkl"), "rb")) for i in range(len(X))]
    idx = np.array([i for i, r in enumerate(res) if r["keep"]])
    out = {"delta": np.array(DELTA), "drawn": np.array(len(X)), "kept_index": idx, "x0": X[idx], "u0": U0[idx], "Q": np.array([res[i]["Q"] for i in idx]),
           "kkt": np.array([res[i]["kkt"] for i in idx]), "n_active": np.array([res[i]["n_active"] for i in idx])}
    for di, d in enumerate(DELTA):
        out[f"dQ_d{di}"] = np.array([res[i][d][0] for i in idx])
        out[f"kkt_d{di}"] = np.array([res[i][d][1] for i in idx])
    np.savez(os.path.join(HERE, "g7d_cartpole_qmode.npz"), **out)
    print("kept", len(idx), "of", len(X))


if __name__ == "__main__":
    main()
