"""Multi-GPU data parallelism for the MPC-as-policy engine: one process per GPU, instances sharded, ONE collective.

The OCP instances of a batch are independent given theta, so the solve itself needs no
communication.  The only exchange of a data-parallel RL update is the accumulated theta-gradient
(rlmpc/examples/linear_system_mpc_qlearning.py:193-205: ``dp = mean_i(lr * td_i * dQ_dp_i)``): every rank
contributes ``[sum_i w_i g_i (n_theta), sum_i w_i, count]`` in fp64 and all ranks apply the identical update.
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``total`` instances over ``world`` ranks (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_weighted_grad(grad: torch.Tensor, weight: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                            deterministic: bool = False, count: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """grad [b, n_theta] (e.g. dQ/dp rows), weight [b] (e.g. lr * td error) of the LOCAL shard.

    Returns (sum_i w_i g_i over ALL ranks [n_theta], sum_i w_i, total count) with a single all-reduce of
    n_theta + 2 doubles; all three are device tensors (no host synchronisation: the caller's next launches queue up
    behind the collective instead of waiting for it).  ``deterministic=True`` sums the local shard with a fixed-shape framework
    reduction (no atomics) and combines the ranks by all-gather + fixed-order summation, so the result is reproducible run to run
    and independent of the collective's reduction tree.  ``count`` [b] (0/1): the count slot of the message carries
    sum(count) instead of the number of rows — the denominator of a mean over the valid samples travels in the SAME message."""
    g = grad.to(torch.float64)
    w = weight.to(torch.float64).reshape(-1)
    n_theta = g.shape[1]
    buf = torch.empty(n_theta + 2, dtype=torch.float64, device=g.device)
    if g.is_cuda and g.stride(1) == 1 and not deterministic:
        # K5 (csrc/reduce_kernel.hpp): one launch instead of a chain of elementwise / reduction kernels
        import ctypes as C
        from . import _lib
        w = w.contiguous()
        # (the library launches on the device that owns `buf`, whatever device is current here)
        rc = _lib.load().mpcrl_weighted_grad_sum(C.c_void_p(g.data_ptr()), int(g.stride(0)), C.c_void_p(w.data_ptr()), int(g.shape[0]),
                                                 int(n_theta), C.c_void_p(buf.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"mpcrl_weighted_grad_sum failed with {rc}")
    else:   # CPU tensors (gloo tests), and the deterministic path: the reduction kernel combines its per-wave partial sums with
        #         floating-point atomics, whose order varies from run to run; a framework reduction of one shape does not
        buf[:n_theta] = (w[:, None] * g).sum(0)
        buf[n_theta] = w.sum()
        buf[n_theta + 1] = float(g.shape[0])
    if count is not None:
        buf[n_theta + 1] = count.to(torch.float64).sum()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if deterministic:
            parts = [torch.empty_like(buf) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, buf, group=group)
            buf = torch.stack(parts).sum(0)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:n_theta], buf[n_theta], buf[n_theta + 1]


def mean_update(grad: torch.Tensor, weight: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                deterministic: bool = False, valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mean_i(w_i * g_i)`` over all instances on all ranks — the parameter step of the Q-learning example
    (linear_system_mpc_qlearning.py:203).  ``valid`` [b] (0/1): samples whose solve failed carry weight 0 and are left out of the
    mean's denominator as well (the reference raises on a failed solve instead, mpc.py:81-83,197-198; a batch cannot)."""
    if valid is None:
        s, _, n = allreduce_weighted_grad(grad, weight, group, deterministic)
    else:
        v = valid.to(torch.float64).reshape(-1)
        # ONE collective per update: the number of valid samples rides in the count slot of the same message
        s, _, n = allreduce_weighted_grad(grad, weight.to(torch.float64).reshape(-1) * v, group, deterministic, count=v)
    return s / torch.clamp(n, min=1.0)
