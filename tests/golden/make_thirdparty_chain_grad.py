"""G8: third-party GRADIENTS of the chain-of-masses NLP.  `python tests/golden/make_thirdparty_chain_grad.py` (dev container, ~1 h).

The reference's own sanity check of dpi/dp on the chain is finite differences along a parameter sweep
(rlmpc/examples/chain_mass.py:28-64, the damping C of one link).  Here scipy's SLSQP — code this repository did not write — solves the
n_mass = 3 NLP (N = 40: 480 unknowns, bounds on the controls) at the perturbed state of G7 and at p (1 +- delta) for four parameters
(the mass m_0, a spring constant D_1_0, a rest length L_0_2 and the swept kind of parameter, the damping C_1_0), delta = 1e-5 and
1e-4; every point is KKT-certified.  Central differences of V and u0* give dV/dp and du0*/dp columns that the port and the HIP path
are held to.  Inputs and expected outputs only (g8_chain_grad.npz).

G8_NMASS=5 (round 6): the size the reference's own test runs (tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174 sweeps
C_3_0 at n_mass 5): 960 unknowns, the state of G7's chain5 rows, the parameters C_3_0 (the swept one) and the mass m_1.  A cold SLSQP
solve of this NLP took 4.4 h (G7) and its full vector was not kept, so the BASE solve starts from the KKT point the C++ port finds
(u0*, V of which equal G7's SLSQP numbers at 1e-6); SLSQP polishes it to its own stopping test and the certificate is computed here.
The base point does not enter a central difference: each gradient entry is the difference of two SLSQP solutions at p (1 +- delta),
which SLSQP reaches from the base point by its own iteration, each KKT-certified."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_thirdparty import Nlp, certify  # noqa: E402
from make_thirdparty_grad import cold, slsqp  # noqa: E402
from oracle.problems import make_chain_mass  # noqa: E402

torch.set_num_threads(1)
# G8_NMASS=4 in the environment: the same for n_mass 4 (720 unknowns), two parameters, written to g8_chain4_grad.npz
N_MASS = int(os.environ.get("G8_NMASS", "3"))
LABELS = {3: ("m_0", "D_1_0", "L_0_2", "C_1_0"), 4: ("D_1_0", "C_2_0"), 5: ("C_3_0", "m_1")}[N_MASS]
DELTA = (1e-5, 1e-4)
PARTS = "/tmp/g8_parts" if N_MASS == 3 else f"/tmp/g8_parts_{N_MASS}"
OUT = "g8_chain_grad.npz" if N_MASS == 3 else f"g8_chain{N_MASS}_grad.npz"


def state():
    from mpc4rl_amd.problems import chain_mass_ocp
    if N_MASS == 5:                          # G7's chain5 state (its generator draws it after the n_mass-3 one from the same stream)
        return chain_mass_ocp(n_mass=5), np.load(os.path.join(HERE, "g7_thirdparty_grad.npz"))["chain5_x0"]
    rng = np.random.default_rng(31)          # the state of G7's chain rows
    ocp = chain_mass_ocp(n_mass=N_MASS)
    M = N_MASS - 2
    x0 = ocp.x0.copy()
    x0[3 * (M + 1):] += rng.normal(0.0, 1e-2, 3 * M)
    return ocp, x0


def job(label):
    import pickle
    f = os.path.join(PARTS, label + ".pkl")
    if os.path.exists(f):
        return label
    ocp, x0 = state()
    P = make_chain_mass(n_mass=N_MASS)
    j = ocp.p_labels.index(label)
    p0 = P.p0.copy()
    nlp = Nlp(P, x0, p0)
    z_start = cold(nlp, x0)
    if N_MASS == 5:
        from oracle import cpu_port
        r = cpu_port.solve(P, x0[None], tol=1e-9, flags=0)
        assert r.status[0] == 0
        z_start = np.concatenate([r.U[0].ravel(), r.X[0, 1:].ravel()])
    zb, vb, itb = slsqp(nlp, z_start, max_rounds=6)
    kb = certify(nlp, zb)
    print("chain", label, "base", zb[: P.nu], vb, itb, kb, flush=True)
    out = {"j": j, "u0": zb[: P.nu].copy(), "V": vb, "kkt": [kb["stationarity"], kb["feasibility"], kb["min_multiplier"]]}
    for d in DELTA:
        pp, pm = p0.copy(), p0.copy()
        pp[j] *= 1.0 + d
        pm[j] *= 1.0 - d
        zp, vp, _ = slsqp(Nlp(P, x0, pp), zb, max_rounds=6)
        zm, vm, _ = slsqp(Nlp(P, x0, pm), zb, max_rounds=6)
        kp, km = certify(Nlp(P, x0, pp), zp), certify(Nlp(P, x0, pm), zm)
        out[d] = ((vp - vm) / (2 * d * p0[j]), (zp[: P.nu] - zm[: P.nu]) / (2 * d * p0[j]), max(kp["stationarity"], km["stationarity"]))
        print("chain", label, d, out[d], flush=True)
    os.makedirs(PARTS, exist_ok=True)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(out, fh)
    os.replace(f + ".tmp", f)
    return label


def main(assemble_only=False):
    import multiprocessing as mp
    import pickle
    os.makedirs(PARTS, exist_ok=True)
    if not assemble_only:
        with mp.get_context("spawn").Pool(len(LABELS)) as pool:
            for lab in pool.imap_unordered(job, LABELS):
                print("done", lab, flush=True)
    ocp, x0 = state()
    res = {lab: pickle.load(open(os.path.join(PARTS, lab + ".pkl"), "rb")) for lab in LABELS if os.path.exists(os.path.join(PARTS, lab + ".pkl"))}
    labs = [lab for lab in LABELS if lab in res]
    assert labs, "no finished job"
    out = {"n_mass": np.array(N_MASS), "x0": x0, "delta": np.array(DELTA), "p_index": np.array([res[lab]["j"] for lab in labs]),
           "p_labels": np.array(labs), "u0": res[labs[0]]["u0"], "V": np.array(res[labs[0]]["V"]),
           "kkt": np.array([res[lab]["kkt"] for lab in labs])}
    for di, d in enumerate(DELTA):
        out[f"dV_d{di}"] = np.array([res[lab][d][0] for lab in labs])
        out[f"du0_d{di}"] = np.array([res[lab][d][1] for lab in labs])          # [param, nu]
        out[f"kkt_d{di}"] = np.array([res[lab][d][2] for lab in labs])
    np.savez(os.path.join(HERE, OUT), **out)
    print("wrote", OUT, "with", labs)


if __name__ == "__main__":
    main(assemble_only=len(sys.argv) > 1 and sys.argv[1] == "assemble")
