"""The oracle's C++ port under AddressSanitizer + UndefinedBehaviorSanitizer ("an ASan build of the CPU twin").

The port is the checker of every GPU parity test; a buffer overrun or an uninitialised read in it would be a wrong reference that still
"agrees with itself".  This builds oracle/cpu with -fsanitize=address,undefined (`make asan`) and runs the three OCPs — solve, dV/dp,
du0*/dp, the RTI step and the exact mode — in a subprocess with the sanitizer runtime preloaded; any report aborts the subprocess
(-fno-sanitize-recover, halt_on_error).  Results are compared with the optimised build: same algorithm, so 1e-9.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import cpu_port
from oracle.problems import make_cartpole, make_chain_mass, make_linear_system
out = {}
rng = np.random.default_rng(5)
P = make_cartpole()
x0 = np.zeros((6, 4)); x0[:, 2] = np.pi + rng.uniform(-0.3, 0.3, 6)
r = cpu_port.solve(P, x0); out["cart"] = np.concatenate([r.u0.ravel(), r.V.ravel(), r.dV.ravel(), r.dpi.ravel()])
r = cpu_port.solve(P, x0, u0fix=np.full((6, 1), 0.1)); out["cartq"] = np.concatenate([r.V.ravel(), r.dV.ravel()])
P = make_linear_system()
x0 = rng.uniform(-0.5, 0.5, (6, 2)); x0[:, 1] = np.abs(x0[:, 1])
r = cpu_port.solve(P, x0); out["lin"] = np.concatenate([r.u0.ravel(), r.V.ravel(), r.dV.ravel(), r.dpi.ravel()])
for n in (3, 4):
    P = make_chain_mass(n_mass=n)
    x0 = np.tile(P.extra["x_ss"], (2, 1)); x0[:, 3 * (n - 3):3 * (n - 2)] += rng.normal(0.0, 1e-2, (2, 3))
    r = cpu_port.solve(P, x0); out["chain%%d" %% n] = np.concatenate([r.u0.ravel(), r.V.ravel(), r.dV.ravel(), r.dpi.ravel()])
np.savez(sys.argv[1], **out)
"""


def _run(env_extra, path):
    env = dict(os.environ, **env_extra)
    subprocess.check_call([sys.executable, "-c", CHILD % {"root": ROOT}, path], env=env, cwd=ROOT)
    return dict(np.load(path))


def test_port_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "cpu"), "asan"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan_rt = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    assert os.path.exists(asan_rt) and os.path.exists(ubsan_rt)
    san = _run({"MPC_ORACLE_LIB": os.path.join(ROOT, "oracle", "_build", "libmpc_oracle_asan.so"), "LD_PRELOAD": asan_rt + ":" + ubsan_rt,
                # (the interpreter itself is not built for leak checking; everything else aborts on the first report)
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "OMP_NUM_THREADS": "2"}, str(tmp_path / "san.npz"))
    ref = _run({"OMP_NUM_THREADS": "2"}, str(tmp_path / "ref.npz"))
    assert set(san) == set(ref) == {"cart", "cartq", "lin", "chain3", "chain4"}
    for k in ref:
        assert np.all(np.isfinite(ref[k]))
        assert np.abs(san[k] - ref[k]).max() <= 1e-9 * max(1.0, np.abs(ref[k]).max()), k
