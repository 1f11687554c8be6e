"""Bit-level A/B of two library builds: a fixed set of solves (all three OCP families, cold / warm / Q-mode / dV-only / per-instance
theta, state-bounded chain) is run with the library MPCRL_LIB_PATH selects and every output is dumped to an .npz; with two files the
script compares them array by array.  A refactor that moves no arithmetic must leave every array bit-identical.

    MPCRL_LIB_PATH=$PWD/build_ab/old.so python profiles/microbench/bitwise_dump.py gpurun_out/old.npz
    python profiles/microbench/bitwise_dump.py gpurun_out/new.npz
    python profiles/microbench/bitwise_dump.py gpurun_out/old.npz gpurun_out/new.npz      # compare
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(path, which):
    import torch
    from mpc4rl_amd import MPCBatch, cartpole_ocp, chain_mass_ocp, linear_system_ocp
    from mpc4rl_amd import _lib
    out = {}

    def put(tag, r, mpc=None):
        torch.cuda.synchronize()
        out[tag + "_u0"], out[tag + "_V"] = r.u0.cpu().numpy(), r.V.cpu().numpy()
        out[tag + "_status"], out[tag + "_iters"] = r.status.cpu().numpy(), r.iters.cpu().numpy()
        if r.dV_dp is not None:
            out[tag + "_dV"] = r.dV_dp.cpu().numpy()
        if r.dpi_dp is not None:
            out[tag + "_dpi"] = r.dpi_dp.cpu().numpy()
        if mpc is not None:
            for nm, t in zip(("x", "u", "pi", "bnd", "res"), mpc.get_iterate()):
                out[tag + "_it_" + nm] = t.cpu().numpy()
            out[tag + "_lag"] = mpc.get_lagrangian().cpu().numpy()

    rng = np.random.default_rng(7)
    if "chain" in which:
        for n_mass, B in ((3, 96), (4, 64), (5, 128), (6, 64), (7, 128)):
            ocp = chain_mass_ocp(n_mass=n_mass)
            M = n_mass - 2
            x0 = np.tile(ocp.x0, (B, 1))
            x0[:, 3 * (M + 1):] += rng.normal(0.0, 1e-2, (B, 3 * M))
            mpc = MPCBatch(ocp, B)
            put(f"chain{n_mass}_cold", mpc.solve(x0, sens_v=True, sens_pi=True, cold=True), mpc)
            x1 = x0 + rng.normal(0.0, 2e-3, x0.shape)
            put(f"chain{n_mass}_warm", mpc.solve(x1, sens_v=True, sens_pi=True), mpc)
            put(f"chain{n_mass}_vonly", mpc.solve(x0, sens_v=True, cold=True))
            u0 = rng.uniform(-0.5, 0.5, (B, 3))
            put(f"chain{n_mass}_q", mpc.solve(x0, u0=u0, sens_v=True, cold=True), mpc)
            th = np.tile(ocp.p0, (B, 1)) * (1.0 + 0.02 * rng.standard_normal((B, ocp.n_p)))
            mpc.set_theta(th)
            put(f"chain{n_mass}_theta", mpc.solve(x0, sens_v=True, sens_pi=True, cold=True))
            if n_mass in (3, 5):      # state bounds on the stages: more than 128 rows -> the workspace form of the row phases
                mpc2 = MPCBatch(ocp, 32)
                nw = ocp.nx + ocp.nu
                lb, ub = np.full(nw, -1e30), np.full(nw, 1e30)
                lb[:3], ub[:3] = -1.0, 1.0
                lb[3:6], ub[3:6] = -5.0, 5.0
                mpc2.set_bounds(_lib.BOUNDS_STAGE, lb, ub)
                put(f"chain{n_mass}_xbnd", mpc2.solve(x0[:32], sens_v=True, sens_pi=True, cold=True), mpc2)
    if "small" in which:
        B = 4096
        x0 = np.zeros((B, 4))
        x0[:, 2] = rng.uniform(0.9 * np.pi, 1.1 * np.pi, B)
        mpc = MPCBatch(cartpole_ocp(), B)
        put("cart_cold", mpc.solve(x0, sens_v=True, sens_pi=True, cold=True), mpc)
        put("cart_warm", mpc.solve(x0 + 0.01 * rng.standard_normal(x0.shape), sens_v=True, sens_pi=True), mpc)
        mpc3 = MPCBatch(cartpole_ocp(), 3072)
        put("cart_plain", mpc3.solve(x0[:3072], sens_v=True, sens_pi=True, cold=True), mpc3)
        put("cart_q", mpc3.solve(x0[:3072], u0=rng.uniform(-10, 10, (3072, 1)), sens_v=True, cold=True))
        xl = np.column_stack([rng.uniform(0.15, 0.85, B), rng.uniform(-0.5, 0.5, B)])
        lin = MPCBatch(linear_system_ocp(discount_factor=0.99), B)
        put("lin_cold", lin.solve(xl, sens_v=True, sens_pi=True, cold=True), lin)
        put("lin_warm", lin.solve(xl + 0.01 * rng.standard_normal(xl.shape), sens_v=True, sens_pi=True), lin)
        put("lin_q", lin.solve(xl, u0=rng.uniform(-0.5, 0.5, (B, 1)), sens_v=True, cold=True))
    np.savez(path, **out)
    print("wrote", path, len(out), "arrays; library", os.environ.get("MPCRL_LIB_PATH", "default"))


def compare(a, b):
    A, Bz = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        if k not in Bz.files:
            print("MISSING", k)
            bad += 1
            continue
        x, y = A[k], Bz[k]
        same = x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
        if not same:
            bad += 1
            d = np.nanmax(np.abs(x.astype(float) - y.astype(float)) / np.maximum(1.0, np.abs(x.astype(float)))) if x.shape == y.shape else -1
            print("DIFF", k, "max rel diff %.3e" % d)
    print("%d arrays, %d differ" % (len(A.files), bad))
    return bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["chain", "small"]
    if len(args) == 2:
        sys.exit(1 if compare(*args) else 0)
    dump(args[0], which)
