// reduce_kernel.hpp — K5: the weighted theta-gradient reduction of a data-parallel RL update on one GPU.
//
//   out[j] = sum_i w_i * g[i, j]   (j < n),   out[n] = sum_i w_i,   out[n + 1] = number of rows
//
// This is the local half of the only exchange on the path (rlmpc/examples/linear_system_mpc_qlearning.py:193-205:
// dp = mean_i(LR * td_i * dQ_dp_i)); the result (n + 2 doubles) is what one RCCL all-reduce then sums over the ranks.
// Two shapes: few columns (cartpole: 3 model parameters) -> rows across lanes, wave reduction, one atomic per wave and column;
// many columns (chain: 499) -> one lane per column, coalesced along the row.
#pragma once
#include <hip/hip_runtime.h>

namespace mpcrl {

constexpr int REDUCE_MAXN_ROWPAR = 16;

__global__ void __launch_bounds__(256) grad_reduce_rows_kernel(const double *g, long ld, const double *w, int rows, int n, double *out) {
    double acc[REDUCE_MAXN_ROWPAR + 1];
#pragma unroll
    for (int j = 0; j <= REDUCE_MAXN_ROWPAR; ++j) acc[j] = 0.0;
    for (long i = blockIdx.x * 256 + threadIdx.x; i < rows; i += (long)gridDim.x * 256) {
        const double wi = w ? w[i] : 1.0;
#pragma unroll
        for (int j = 0; j < REDUCE_MAXN_ROWPAR; ++j)
            if (j < n) acc[j] = fma(wi, g[i * ld + j], acc[j]);
        acc[REDUCE_MAXN_ROWPAR] += wi;
    }
#pragma unroll
    for (int j = 0; j <= REDUCE_MAXN_ROWPAR; ++j) {
        if (j < n || j == REDUCE_MAXN_ROWPAR) {
            double v = acc[j];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
            if ((threadIdx.x & 63) == 0) atomicAdd(out + (j == REDUCE_MAXN_ROWPAR ? n : j), v);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n + 1] = (double)rows;
}

__global__ void __launch_bounds__(256) grad_reduce_cols_kernel(const double *g, long ld, const double *w, int rows, int n, double *out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    // blockIdx.y splits the rows; partial sums meet in out through atomics
    const int r0 = (int)((long)rows * blockIdx.y / gridDim.y), r1 = (int)((long)rows * (blockIdx.y + 1) / gridDim.y);
    double acc = 0.0, wsum = 0.0;
    for (int i = r0; i < r1; ++i) {
        const double wi = w ? w[i] : 1.0;
        if (j < n) acc = fma(wi, g[(long)i * ld + j], acc);
        wsum += wi;
    }
    if (j < n) atomicAdd(out + j, acc);
    if (j == 0) atomicAdd(out + n, wsum);
    if (j == 0 && blockIdx.y == 0) out[n + 1] = (double)rows;
}

}  // namespace mpcrl
