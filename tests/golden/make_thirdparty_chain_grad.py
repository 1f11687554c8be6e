"""G8: third-party GRADIENTS of the chain-of-masses NLP.  `python tests/golden/make_thirdparty_chain_grad.py` (dev container, ~1 h).

The reference's own sanity check of dpi/dp on the chain is finite differences along a parameter sweep
(rlmpc/examples/chain_mass.py:28-64, the damping C of one link).  Here scipy's SLSQP — code this repository did not write — solves the
n_mass = 3 NLP (N = 40: 480 unknowns, bounds on the controls) at the perturbed state of G7 and at p (1 +- delta) for four parameters
(the mass m_0, a spring constant D_1_0, a rest length L_0_2 and the swept kind of parameter, the damping C_1_0), delta = 1e-5 and
1e-4; every point is KKT-certified.  Central differences of V and u0* give dV/dp and du0*/dp columns that the port and the HIP path
are held to.  Inputs and expected outputs only (g8_chain_grad.npz).

G8_NMASS=5 (round 6): the size the reference's own test runs (tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174 sweeps
C_3_0 at n_mass 5): 960 unknowns, the state of G7's chain5 rows, the parameters C_3_0 (the swept one) and the mass m_1.  SLSQP needs
2.6 s per iteration and thousands of iterations per solve at this size (G7's ONE cold solve: 6000 iterations, 4.4 h; its full vector was
not kept; five more solves per parameter do not fit a session), so the points are found by another third-party solver: MINPACK's
Powell hybrid method (scipy.optimize.root, method "hybr", analytic Jacobian) on the KKT EQUATIONS of the NLP with the active set of
the solution — stationarity with multipliers for the dynamics and the active control bounds, the dynamics, the active bounds — started
at p from the KKT point the C++ port finds (its u0*, V equal G7's SLSQP numbers at 1e-6, asserted below) and at p (1 +- delta) from
the solution at p.  Objective, dynamics, their derivatives: the NLP restatement of make_thirdparty.Nlp through torch autograd (the
Hessian of the Lagrangian too); nothing of oracle/'s or the kernels' derivative or sensitivity code.  Every point is certified HERE as
all the others are (make_thirdparty.certify: multipliers by least squares, stationarity, feasibility, signs, bounds), so that the
active set assumed is the active set of a KKT point.  The base point does not enter a central difference."""
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
LABELS = {3: ("m_0", "D_1_0", "L_0_2", "C_1_0"), 4: ("D_1_0", "C_2_0"),
          5: ("C_3_0", "m_1", "D_2_1", "L_1_2", "C_0_2", "m_3", "Q_44", "R_4"), 7: ("C_5_0", "m_2", "D_3_1", "L_0_2")}[N_MASS]
# G8_SWEEP=1 (with G8_NMASS=5): the reference's own test instead — tests/test_chain_mass.py -> rlmpc/examples/chain_mass.py:133-174: the
# unperturbed x0 of examples/chain_mass.py:17-25, C_3_0 over linspace(0.05, 0.15, 10), pi* and d pi / d C_3_0 at every point
SWEEP = bool(os.environ.get("G8_SWEEP")) and N_MASS == 5
SWEEP_VALUES = np.linspace(0.05, 0.15, 10)
DELTA = (1e-5, 1e-4)
PARTS = "/tmp/g8_parts" if N_MASS == 3 else (f"/tmp/g8_parts_{N_MASS}" if N_MASS not in (5, 7) else f"/tmp/g8_parts_{N_MASS}k" + ("s" if SWEEP else ""))
OUT = "g8_chain_grad.npz" if N_MASS == 3 else (f"g8_chain{N_MASS}_grad.npz" if not SWEEP else "g8_chain5_sweep.npz")
if SWEEP:
    LABELS = tuple(f"sweep{i}" for i in range(len(SWEEP_VALUES)))


def state():
    from mpc4rl_amd.problems import chain_mass_ocp
    if SWEEP:
        ocp = chain_mass_ocp(n_mass=5)
        return ocp, ocp.x0.copy()
    if N_MASS == 5:                          # G7's chain5 state (its generator draws it after the n_mass-3 one from the same stream)
        return chain_mass_ocp(n_mass=5), np.load(os.path.join(HERE, "g7_thirdparty_grad.npz"))["chain5_x0"]
    rng = np.random.default_rng(31)          # the state of G7's chain rows
    ocp = chain_mass_ocp(n_mass=N_MASS)
    M = N_MASS - 2
    x0 = ocp.x0.copy()
    x0[3 * (M + 1):] += rng.normal(0.0, 1e-2, 3 * M)
    return ocp, x0


def kkt_root(nlp, z_start, w_start=None, act_tol=1e-7):
    """KKT point of the NLP on the active set of z_start by scipy.optimize.root (MINPACK hybrj): unknowns w = [z; nu; mu_A]."""
    from scipy.optimize import root
    n = nlp.nz
    lo_a, hi_a = np.where(z_start - nlp.lo < act_tol)[0], np.where((nlp.hi - z_start < act_tol) & (nlp.hi > nlp.lo))[0]      # (a pinned variable, lo = hi, is ONE equality row)
    act = np.concatenate([lo_a, hi_a])
    bval = np.concatenate([nlp.lo[lo_a], nlp.hi[hi_a]])
    m = nlp.P.N * nlp.P.nx
    E = np.zeros((len(act), n))
    E[np.arange(len(act)), act] = 1.0
    t = lambda a: torch.tensor(a, dtype=torch.float64)

    def F(w):
        z, nu, mu = w[:n], w[n:n + m], w[n + m:]
        zt = t(z)
        return np.concatenate([nlp.gradf(zt).numpy() + nlp.Jg(zt).numpy().T @ nu + E.T @ mu, nlp.g(zt).numpy(), z[act] - bval])

    def J(w):
        z, nu = w[:n], t(w[n:n + m])
        zt = t(z)
        H = torch.autograd.functional.hessian(lambda zz: nlp.f(zz) + nu @ nlp.g(zz), zt, vectorize=True).numpy()
        Jg = nlp.Jg(zt).numpy()
        k = len(act)
        return np.block([[H, Jg.T, E.T], [Jg, np.zeros((m, m)), np.zeros((m, k))], [E, np.zeros((k, m)), np.zeros((k, k))]])

    if w_start is None:      # multipliers of the starting point by least squares on its stationarity rows
        zt = t(z_start)
        A = np.column_stack([nlp.Jg(zt).numpy().T, E.T])
        mult = np.linalg.lstsq(A, -nlp.gradf(zt).numpy(), rcond=None)[0]
        w_start = np.concatenate([z_start, mult])
    r = root(F, w_start, jac=J, method="hybr", tol=1e-15, options={"xtol": 1e-15, "maxfev": 200})
    w = r.x
    res = float(np.abs(F(w)).max())
    return w[:n].copy(), float(nlp.f(t(w[:n]))), w, res, int(r.nfev)


def job(label):
    if N_MASS in (5, 7):
        return job_kkt(label)
    return job_slsqp(label)


def job_kkt(label):
    import pickle
    from oracle import cpu_port
    f = os.path.join(PARTS, label + ".pkl")
    if os.path.exists(f):
        return label
    ocp, x0 = state()
    P = make_chain_mass(n_mass=N_MASS)
    p0 = P.p0.copy()
    if SWEEP:
        j = ocp.p_labels.index("C_3_0")
        p0[j] = SWEEP_VALUES[int(label[5:])]
    else:
        j = ocp.p_labels.index(label)
    nlp = Nlp(P, x0, p0)
    r = cpu_port.solve(P, x0[None], p=p0, tol=1e-9, flags=0)
    assert r.status[0] == 0
    zb, vb, wb, resb, nfb = kkt_root(nlp, np.concatenate([r.U[0].ravel(), r.X[0, 1:].ravel()]))
    kb = certify(nlp, zb)
    if N_MASS == 5 and not SWEEP:
        g7 = np.load(os.path.join(HERE, "g7_thirdparty_grad.npz"))       # SLSQP's cold solve of the same NLP: the same KKT point
        assert np.abs(zb[: P.nu] - g7["chain5_u0"]).max() < 1e-6 and abs(vb - float(g7["chain5_V"])) < 1e-6 * abs(vb), (zb[: P.nu], vb)
    print("chain", label, "base", zb[: P.nu], vb, "KKT residual", resb, "evaluations", nfb, kb, flush=True)
    out = {"j": j, "u0": zb[: P.nu].copy(), "V": vb, "kkt": [kb["stationarity"], kb["feasibility"], kb["min_multiplier"]]}
    for d in DELTA:
        pp, pm = p0.copy(), p0.copy()
        pp[j] *= 1.0 + d
        pm[j] *= 1.0 - d
        zp, vp, _, rp, _ = kkt_root(Nlp(P, x0, pp), zb, wb)
        zm, vm, _, rm, _ = kkt_root(Nlp(P, x0, pm), zb, wb)
        kp, km = certify(Nlp(P, x0, pp), zp), certify(Nlp(P, x0, pm), zm)
        assert kp["n_active"] == kb["n_active"] == km["n_active"] and min(kp["min_multiplier"], km["min_multiplier"]) > 0.0
        out[d] = ((vp - vm) / (2 * d * p0[j]), (zp[: P.nu] - zm[: P.nu]) / (2 * d * p0[j]), max(kp["stationarity"], km["stationarity"]))
        print("chain", label, d, out[d], "KKT residuals", rp, rm, flush=True)
    os.makedirs(PARTS, exist_ok=True)
    with open(f + ".tmp", "wb") as fh:
        pickle.dump(out, fh)
    os.replace(f + ".tmp", f)
    return label


def job_slsqp(label):
    import pickle
    f = os.path.join(PARTS, label + ".pkl")
    if os.path.exists(f):
        return label
    ocp, x0 = state()
    P = make_chain_mass(n_mass=N_MASS)
    j = ocp.p_labels.index(label)
    p0 = P.p0.copy()
    nlp = Nlp(P, x0, p0)
    zb, vb, itb = slsqp(nlp, cold(nlp, x0), max_rounds=6)
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
        with mp.get_context("spawn").Pool(min(len(LABELS), int(os.environ.get("G8_PROCS", "4")))) as pool:
            for lab in pool.imap_unordered(job, LABELS):
                print("done", lab, flush=True)
    ocp, x0 = state()
    res = {lab: pickle.load(open(os.path.join(PARTS, lab + ".pkl"), "rb")) for lab in LABELS if os.path.exists(os.path.join(PARTS, lab + ".pkl"))}
    labs = [lab for lab in LABELS if lab in res]
    assert labs, "no finished job"
    out = {"n_mass": np.array(N_MASS), "x0": x0, "delta": np.array(DELTA), "p_index": np.array([res[lab]["j"] for lab in labs]),
           "p_labels": np.array(labs), "u0": res[labs[0]]["u0"], "V": np.array(res[labs[0]]["V"]),
           "kkt": np.array([res[lab]["kkt"] for lab in labs])}
    if SWEEP:      # one base point per swept value
        out.update(C_3_0=SWEEP_VALUES[[int(lab[5:]) for lab in labs]], u0=np.array([res[lab]["u0"] for lab in labs]), V=np.array([res[lab]["V"] for lab in labs]))
    for di, d in enumerate(DELTA):
        out[f"dV_d{di}"] = np.array([res[lab][d][0] for lab in labs])
        out[f"du0_d{di}"] = np.array([res[lab][d][1] for lab in labs])          # [param, nu]
        out[f"kkt_d{di}"] = np.array([res[lab][d][2] for lab in labs])
    np.savez(os.path.join(HERE, OUT), **out)
    print("wrote", OUT, "with", labs)


if __name__ == "__main__":
    main(assemble_only=len(sys.argv) > 1 and sys.argv[1] == "assemble")
