// iterate_kernel.hpp — the cold iterate of MPC.reset (rlmpc/mpc/common/mpc.py:204-210) written into the handle's stored-iterate arrays:
// x_k = x0 for every stage (or 0 when no x0 is given), u = 0, pi = 0, bound multipliers 0 and slacks t = 1 (the state MPCRL_COLD builds
// inside the solve kernels).  Grid-stride over the largest array; any of X / U / PI / BND may be null (left untouched).
#pragma once
#include <hip/hip_runtime.h>

namespace mpcrl {

__global__ void cold_iterate_kernel(const double *x0, int B, int N, int nx, int nu, double *X, double *U, double *PI, double *BND) {
    const long nw = nx + nu, nb = (long)(N + 1) * nw;
    const long nX = (long)B * (N + 1) * nx, nU = (long)B * N * nu, nP = (long)B * N * nx, nB = (long)B * 10 * nb;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nB; i += stride) {
        if (BND) {
            const long j = (i / nb) % 10;   // plane: lam_l, lam_u, t_l, t_u, s_l, s_u, lam_sl, lam_su, t_sl, t_su
            BND[i] = (j == 2 || j == 3 || j == 8 || j == 9) ? 1.0 : 0.0;
        }
        if (X && i < nX) X[i] = x0 ? x0[(i / ((long)(N + 1) * nx)) * nx + i % nx] : 0.0;
        if (U && i < nU) U[i] = 0.0;
        if (PI && i < nP) PI[i] = 0.0;
    }
}

// Row-indexed moves of whole iterates between the handle's stored-iterate arrays (H*, one row per instance) and caller-owned tables laid
// out the same way with any number of rows (a replay buffer of solver iterates: mpcrl_get_iterate_rows / mpcrl_set_iterate_rows).
// TO_HANDLE: H[i] = T[index[i]]; else T[index[i]] = H[i]; index[i] < 0 skips instance i.  One 256-lane workgroup per instance, the four arrays back to back, 8-byte
// elements, consecutive lanes on consecutive elements (a cartpole iterate is 9.9 KB: 40 MB per 4096 instances, one pass each way).
template <bool TO_HANDLE>
__global__ void __launch_bounds__(256) iterate_rows_kernel(double *HX, double *HU, double *HPI, double *HB, double *TX, double *TU, double *TPI, double *TB,
                                                           const long *index, int nX, int nU, int nP, int nB) {
    const long i = blockIdx.x, r = index ? index[i] : i;
    if (r < 0) return;      // (a negative row: this instance is left alone / not recorded)
    auto move = [&](double *H, double *T, int n) {
        if (!H || !T) return;
        double *h = H + i * n, *t = T + r * n;
        for (int e = threadIdx.x; e < n; e += 256) {
            if (TO_HANDLE) h[e] = t[e]; else t[e] = h[e];
        }
    };
    move(HX, TX, nX), move(HU, TU, nU), move(HPI, TPI, nP), move(HB, TB, nB);
}

}  // namespace mpcrl
