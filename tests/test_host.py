"""Host-side logic: problem descriptors restate the reference's problem data; sharding helpers."""
import numpy as np


def test_cartpole_descriptor_matches_reference_layout():
    from mpc4rl_amd import cartpole_ocp
    o = cartpole_ocp()
    assert (o.nx, o.nu, o.N, o.n_p, o.n_model_p) == (4, 1, 20, 83, 3)     # n_p = 3 + 25 + 25 + 16 + 5 + 5 + 4
    assert abs(o.dT - 0.1) < 1e-15 and abs(o.h - 0.025) < 1e-15            # dt = tf/N, h = dt/4 (cartpole/acados.py:86-92)
    assert np.allclose(o.p0[:3], [1.0, 0.1, 0.5])                          # config/cartpole.yaml:63-74
    assert np.allclose(np.diag(o.p0[3:28].reshape(5, 5)), [10, 0.1, 10, 0.1, 0.01])
    lb, ub, lbe, ube, soft, zl, zu = o.stage_bounds()
    assert np.allclose(ub, [30, 2.4, 10, 6.28, 10]) and np.allclose(lb, -ub) and np.allclose(ube, ub[1:])
    # n_h of the mirror: stage 0: 2 nu + 2 nx, middle: 2 (nu + nx), terminal 2 nx  -> 208
    assert (2 * 1 + 2 * 4) + 19 * 10 + 8 == 208


def test_linear_descriptor_matches_reference_layout():
    from mpc4rl_amd import linear_system_ocp
    o = linear_system_ocp()
    assert (o.nx, o.nu, o.N, o.n_p) == (2, 1, 40, 12)
    assert np.allclose(o.p0, [1, 0, 0.25, 1, 0.03125, 0.25, 0, 0, 1e-3, 0, 0, 0])   # vec(A) column-major, B, b, V_0, f
    P = o.consts.reshape(2, 2)
    assert np.allclose(P, [[7.464124567610356, 4.031128874149254], [4.031128874149254, 7.518320906916593]])
    lb, ub, lbe, ube, soft, zl, zu = o.stage_bounds()
    assert list(soft) == [0, 1, 0] and zl[1] == 100.0 and zu[1] == 100.0
    assert np.all(np.abs(lbe) >= 1e29)                                     # no terminal bound


def test_descriptors_agree_with_oracle_problem_data():
    """Two independent restatements of the reference's numbers (product vs oracle) must coincide."""
    from mpc4rl_amd import cartpole_ocp, linear_system_ocp
    from oracle.cpu_port import stage_bounds
    from oracle.problems import make_cartpole, make_linear_system
    for o, P in ((cartpole_ocp(), make_cartpole()), (linear_system_ocp(), make_linear_system())):
        assert np.allclose(o.p0, P.p0) and o.N == P.N and o.dT == P.dT and o.p_labels == P.p_labels
        for a, b in zip(o.stage_bounds(), stage_bounds(P)):
            assert np.array_equal(a, b)


def test_shard_range_covers_batch():
    from mpc4rl_amd.distributed import shard_range
    for total, world in ((4096, 8), (10, 3), (5, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
