"""CPU oracle for the MPC-as-policy hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, link or execute it, and there only as the checker / reported CPU
baseline, never as the thing measured or shipped.  The product path
(``mpc4rl_amd``) fails loudly when its HIP library is missing; it never
falls back to this code.

PARITY STATUS: **parity unpinned at the acados boundary.**  The arithmetic of
the reference's solve lives in acados/HPIPM/BLASFEO and CasADi, which are not
vendored in the reference tree (``.gitignore:478``), not pinned
(``pyproject.toml:8-15``) and not installable here.  The reference's tests hold
no golden numbers (``tests/test_linear_example.py:8-26`` asserts construction
only; ``tests/test_chain_mass.py:4-6`` ends in ``assert True``).  What pins this
oracle instead:

* G1  closed-form LQR/DARE known answers (``tests/golden/g1_lqr.json``);
* the KKT-consistency thresholds the reference applies to every solver
  output inside ``update_nlp`` (``rlmpc/mpc/nlp.py:1445-1537``), applied here
  to the oracle's own solution;
* central finite differences of V, Q and u0* over the parameters (the pattern
  of ``scripts/linear_system_mpc_nlp.py:17-106``);
* an *independent derivative path*: ``nlp_mirror`` differentiates the residual
  ``R(z, p)`` of ``rlmpc/mpc/nlp.py:1214`` with torch autograd and solves the
  dense system of ``nlp.py:1410-1424``, whereas the engine uses hand-written
  forward-mode derivatives and an adjoint Riccati sweep.

Layout
------
problems.py     problem data of the three reference OCPs (torch float64)
sqp_dense.py    dense full-step SQP + dense primal-dual IPM (numpy)      "O2"
nlp_mirror.py   restatement of build_nlp/update_nlp (L, R, z, p layouts)  "O1"
cpu/            C++ port of the structured algorithm (CPU baseline, "port")
"""
