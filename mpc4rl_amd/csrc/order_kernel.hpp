// order_kernel.hpp — packing order of a batch (scheduling hint, no effect on results).
//
// Instances that share a wavefront run in lock-step for the maximum of their iteration counts, so neighbours should be similar
// problems: the batch is ordered along the coordinate of x0 with the largest spread.  Exact order is not needed, similar
// neighbours are: one workgroup does a counting sort into 1024 buckets of that coordinate (LDS atomics, block scan) and then
// orders every bucket by instance index, which makes the permutation deterministic.  ~15 us for 4096 instances, instead of
// ~130 us for an argsort expressed as a chain of framework kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace mpcrl {

constexpr int ORDER_NT = 1024, ORDER_MAX = 8192, ORDER_PER = ORDER_MAX / ORDER_NT, ORDER_BUCKET_MAX = 64;

// state[3] = {spread, minimum, coordinate} of the batch the order was last built from scratch for (refresh != 0): finding the
// coordinate with the largest spread is four dependent passes over x0 — 7 of the kernel's 11 us — and consecutive batches of a
// handle are alike, so in between the kernel buckets along the remembered coordinate and range (keys outside it land in the end
// buckets; a batch that no longer spreads along it ends in one crowded bucket = the identity order below).
__global__ void __launch_bounds__(ORDER_NT) order_kernel(const double *x0, int B, int nx, int *perm, double *state, int refresh) {
    __shared__ int cnt[ORDER_NT], off[ORDER_NT], wsum[ORDER_NT / 64], tmp[ORDER_MAX];
    __shared__ double red[2 * (ORDER_NT / 64)];
    __shared__ double best[3];   // spread, min, dim
    const int tid = threadIdx.x;
    if (tid == 0) best[0] = -1.0, best[1] = 0.0, best[2] = 0.0;
    cnt[tid] = 0;
    __syncthreads();
    if (refresh) {
        for (int d = 0; d < nx; ++d) {
            double lo = 1e300, hi = -1e300;
            for (int i = tid; i < B; i += ORDER_NT) {
                const double v = x0[(size_t)i * nx + d];
                if (v == v) lo = fmin(lo, v), hi = fmax(hi, v);   // NaN inputs do not take part
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) lo = fmin(lo, __shfl_xor(lo, s)), hi = fmax(hi, __shfl_xor(hi, s));
            if ((tid & 63) == 0) red[2 * (tid >> 6)] = lo, red[2 * (tid >> 6) + 1] = hi;
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < ORDER_NT / 64; ++w) lo = fmin(lo, red[2 * w]), hi = fmax(hi, red[2 * w + 1]);
                if (hi - lo > best[0]) best[0] = hi - lo, best[1] = lo, best[2] = (double)d;
            }
            __syncthreads();
        }
        if (tid == 0) state[0] = best[0], state[1] = best[1], state[2] = best[2];
    } else {
        if (tid == 0) best[0] = state[0], best[1] = state[1], best[2] = state[2];
        __syncthreads();
    }
    const int dim = min(max((int)best[2], 0), nx - 1);   // (a state nobody refreshed holds {-1, 0, 0}: one crowded bucket, identity order)
    const double lo = best[1], scale = best[0] > 0.0 ? (double)ORDER_NT / best[0] : 0.0;
    int bkt[ORDER_PER], rnk[ORDER_PER];
#pragma unroll
    for (int e = 0; e < ORDER_PER; ++e) {
        const int i = tid + e * ORDER_NT;
        bkt[e] = 0, rnk[e] = 0;
        if (i < B) {
            const double v = x0[(size_t)i * nx + dim];
            const double q = (v == v) ? fmin(fmax((v - lo) * scale, 0.0), (double)(ORDER_NT - 1)) : (double)(ORDER_NT - 1);   // NaN last
            bkt[e] = (int)q;
            rnk[e] = atomicAdd(&cnt[bkt[e]], 1);
        }
    }
    __syncthreads();
    {   // A heavily populated bucket means the batch is clustered along the chosen coordinate (all environments reset to one state,
        // say): its members are similar problems whatever their order, and ordering it by index below would be a serial insertion
        // sort of up to B entries on one thread.  Then the identity permutation is the packing order.
        __shared__ int crowded;
        if (tid == 0) crowded = 0;
        __syncthreads();
        if (cnt[tid] > ORDER_BUCKET_MAX) crowded = 1;
        __syncthreads();
        if (crowded) {
            for (int i = tid; i < B; i += ORDER_NT) perm[i] = i;
            return;
        }
    }
    {   // exclusive scan of the bucket counts
        const int c = cnt[tid];
        int incl = c;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_up(incl, s);
            if ((tid & 63) >= s) incl += o;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
        off[tid] = base + incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < ORDER_PER; ++e) {
        const int i = tid + e * ORDER_NT;
        if (i < B) tmp[off[bkt[e]] + rnk[e]] = i;
    }
    __syncthreads();
    {   // the atomics filled each bucket in arrival order: order it by index (insertion sort, a handful of entries)
        const int o = off[tid], n = cnt[tid];
        for (int a = 1; a < n; ++a) {
            const int v = tmp[o + a];
            int b = a - 1;
            while (b >= 0 && tmp[o + b] > v) tmp[o + b + 1] = tmp[o + b], --b;
            tmp[o + b + 1] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < B; i += ORDER_NT) perm[i] = tmp[i];
}

}  // namespace mpcrl
