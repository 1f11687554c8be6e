// mall_probe.hip — does a per-wavefront block set that is re-read in alternating directions (what the Riccati sweeps of the chain solver
// do with the stage blocks of one instance) come out of the Infinity Cache?  1024 wavefronts (one per SIMD), each owns `nblk` blocks
// of `blk` bytes and sweeps over them `reps` times, forward then backward, 4 blocks in flight; reports GB/s against the footprint.
//   hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NREG>   // NREG doubles per lane and block: block = NREG * 512 bytes
__global__ void __launch_bounds__(64, 1) sweep(const double *base, size_t stride, int nblk, int reps, double *out) {
    const double *p = base + (size_t)blockIdx.x * stride + threadIdx.x;
    double acc[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) acc[r] = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        const bool fwd = (rep & 1) == 0;
        for (int i = 0; i < nblk; i += 4) {
            double v[4][NREG];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int k = fwd ? i + d : nblk - 1 - (i + d);
#pragma unroll
                for (int r = 0; r < NREG; ++r) v[d][r] = p[((size_t)k * NREG + r) * 64];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc[r] += v[d][r];
        }
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) s += acc[r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    const int W = 1024, reps = 40;
    double *buf, *out;
    const size_t maxbytes = (size_t)1 << 30;
    hipMalloc(&buf, maxbytes);
    hipMalloc(&out, W * 64 * 8);
    hipMemset(buf, 0, maxbytes);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int nblk : {4, 8, 16, 24, 32, 40, 48, 64, 96, 128, 192}) {
        constexpr int NREG = 9;   // 4.6 KB blocks (n_mass 5)
        const size_t stride = (size_t)nblk * NREG * 64;
        if (stride * 8 * W > maxbytes) break;
        sweep<NREG><<<W, 64>>>(buf, stride, nblk, 2, out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        sweep<NREG><<<W, 64>>>(buf, stride, nblk, reps, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)W * reps * nblk * NREG * 512;
        printf("blocks %3d footprint %7.1f MB  %8.3f ms  %8.1f GB/s\n", nblk, W * stride * 8 / 1e6, ms, bytes / ms / 1e6);
    }
    return 0;
}
