"""world_size-2 gloo test (CPU) of the only collective on the path: the theta-gradient all-reduce of a data-parallel
RL update.  The per-instance quantities come from the CPU oracle port here (no GPU in this test);
the thing under test is the sharding + reduction logic of mpc4rl_amd.distributed."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, grads, weights, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpc4rl_amd.distributed import allreduce_weighted_grad, mean_update, shard_range
    lo, hi = shard_range(grads.shape[0], rank, world)
    g, w = torch.as_tensor(grads[lo:hi]), torch.as_tensor(weights[lo:hi])
    s, ws, n = allreduce_weighted_grad(g, w)
    s_det, _, _ = allreduce_weighted_grad(g, w, deterministic=True)
    step = mean_update(g, w)
    # mean over the valid samples: the denominator rides in the SAME message (one collective per update)
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    valid = torch.as_tensor((np.arange(lo, hi) % 3 != 0).astype(np.float64))
    step_v = mean_update(g, w, valid=valid)
    dist.all_reduce = real
    out[rank] = (s.numpy().copy(), float(ws), int(round(float(n))), s_det.numpy().copy(), step.numpy().copy(), step_v.numpy().copy(), len(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_theta_gradient_allreduce_world2(oracle_port):
    from oracle.problems import make_linear_system
    P = make_linear_system(gamma=0.9)
    rng = np.random.default_rng(0)
    B = 10
    x0 = np.column_stack([rng.uniform(0.2, 0.8, B), rng.uniform(-0.4, 0.4, B)])
    u0 = rng.uniform(-0.8, 0.8, (B, 1))
    q = oracle_port.solve(P, x0, u0fix=u0)                      # dQ/dp per replay sample (q_update, mpc.py:52-96)
    td = rng.normal(size=B)                                     # stand-in TD errors
    weights = 1e-4 * td                                         # LR * td (linear_system_mpc_qlearning.py:203)
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), q.dV, weights, out), nprocs=world, join=True)
    expect_sum = (weights[:, None] * q.dV).sum(0)
    vmask = (np.arange(B) % 3 != 0).astype(float)
    expect_v = ((weights * vmask)[:, None] * q.dV).sum(0) / vmask.sum()
    for r in range(world):
        s, ws, n, s_det, step, step_v, n_coll = out[r]
        assert n_coll == 1, "mean_update(valid=...) must issue exactly one collective"
        assert np.allclose(step_v, expect_v, rtol=1e-12, atol=1e-15)
        assert n == B and abs(ws - weights.sum()) < 1e-15
        assert np.allclose(s, expect_sum, rtol=1e-12, atol=1e-15)
        assert np.allclose(s_det, expect_sum, rtol=1e-12, atol=1e-15)
        assert np.allclose(step, expect_sum / B, rtol=1e-12, atol=1e-15)
    assert np.array_equal(out[0][3], out[1][3])                  # deterministic path: bitwise identical on all ranks
    assert np.array_equal(out[0][4], out[1][4])                  # every rank applies the identical parameter step
