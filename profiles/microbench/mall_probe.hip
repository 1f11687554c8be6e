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

// the same sweep with the access shape of the lane-per-column vector sweeps: NWD active lanes, NWD load instructions of NWD x 8 bytes per block
template <int NWD, int DEPTH, int NSTORE = 0>
__global__ void __launch_bounds__(64, 1) sweep_rows(const double *base, size_t stride, int nblk, int reps, double *out, double *vec = nullptr) {
    const int lane = threadIdx.x < NWD ? threadIdx.x : NWD - 1;
    const double *p = base + (size_t)blockIdx.x * stride + lane;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rep = 0; rep < reps; ++rep) {
        const bool fwd = (rep & 1) == 0;
        for (int i = 0; i < nblk; i += DEPTH) {
            double v[DEPTH][NWD];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int k = fwd ? i + d : nblk - 1 - (i + d);
#pragma unroll
                for (int r = 0; r < NWD; ++r) v[d][r] = p[((size_t)k * NWD + r) * NWD];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int r = 0; r < NWD; ++r) acc[r & 3] += v[d][r];
                if constexpr (NSTORE > 0) {      // the sweeps' result stores: NSTORE instructions of 4 lanes x 8 bytes per stage, natural-order vectors
                    const int k = fwd ? i + d : nblk - 1 - (i + d);
                    if ((threadIdx.x & 15) == 0)
#pragma unroll
                        for (int s_ = 0; s_ < NSTORE; ++s_) vec[(size_t)blockIdx.x * 4096 + (size_t)k * 4 * NSTORE + 4 * s_ + (threadIdx.x >> 4)] = acc[s_ & 3];
                }
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int NWD, int DEPTH>
void run_rows(double *buf, double *out, hipEvent_t a, hipEvent_t b, int W, int reps, size_t stride_bytes = 0) {
    const int nblk = 40 / DEPTH * DEPTH;
    const size_t stride = stride_bytes ? stride_bytes / 8 : (size_t)40 * NWD * NWD;
    sweep_rows<NWD, DEPTH><<<W, 64>>>(buf, stride, nblk, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    sweep_rows<NWD, DEPTH><<<W, 64>>>(buf, stride, nblk, reps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)W * reps * nblk * NWD * NWD * 8;
    printf("rows: NW %d depth %d waves %4d footprint %7.1f MB  %8.3f ms  %8.1f GB/s\n", NWD, DEPTH, W, W * stride * 8 / 1e6, ms, bytes / ms / 1e6);
}

int main() {
    const int W = 1024, reps = 40;
    double *buf, *out;
    const size_t maxbytes = (size_t)3 << 30;
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
    // the same streams spread out: one instance's workspace is ~1.3 - 3 MB, of which the sweeps read one 184 / 415 KB array
    for (size_t st : {(size_t)512 << 10, (size_t)1 << 20, (size_t)3 << 19, (size_t)5 << 19}) {
        printf("stride %zu KB: ", st >> 10);
        run_rows<24, 4>(buf, out, a, b, 1024, reps, st);
        printf("stride %zu KB: ", st >> 10);
        run_rows<36, 2>(buf, out, a, b, 1024, reps, st);
    }
    {   // with the small result stores of the sweeps
        double *vec;
        hipMalloc(&vec, (size_t)1024 * 4096 * 8);
        const int nblk = 40;
        const size_t stride = (size_t)40 * 24 * 24;
        auto go = [&](auto kern, const char *what) {
            kern<<<1024, 64>>>(buf, stride, nblk, 2, out, vec);
            hipDeviceSynchronize();
            hipEventRecord(a);
            kern<<<1024, 64>>>(buf, stride, nblk, reps, out, vec);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("%s: %8.3f ms  %8.1f GB/s\n", what, ms, (double)1024 * reps * nblk * 24 * 24 * 8 / ms / 1e6);
        };
        go(sweep_rows<24, 4, 0>, "rows NW 24 depth 4, no stores      ");
        go(sweep_rows<24, 4, 1>, "rows NW 24 depth 4, 1 small store  ");
        go(sweep_rows<24, 4, 7>, "rows NW 24 depth 4, 7 small stores ");
    }
    for (int w : {1024, 512, 256}) {
        run_rows<24, 2>(buf, out, a, b, w, reps);
        run_rows<24, 4>(buf, out, a, b, w, reps);
        run_rows<36, 2>(buf, out, a, b, w, reps);
    }
    return 0;
}
