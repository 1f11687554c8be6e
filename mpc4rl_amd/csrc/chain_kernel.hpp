// chain_kernel.hpp — the chain-of-masses solver (nx = 9 .. 33, nu = 3; rlmpc/mpc/chain_mass/ocp_utils.py:59-147,195-316), in four parts:
//   chain_common.hpp     data layout (workspace, Omega coordinates, LDS plan), wavefront-level building blocks
//   chain_sweeps.hpp     the QP: interior-point iteration, Riccati sweeps as register-resident MFMA pipelines, phase calls
//   chain_linearise.hpp  iterate set-up, linearisation passes, the SQP kernel
//   chain_sens.hpp       dV/dp and du0*/dp
#pragma once
#include "chain_sens.hpp"
