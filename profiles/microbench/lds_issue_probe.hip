// lds_issue_probe — what an LDS instruction costs the ISSUING wavefront when nothing hides it (gfx950, one wavefront per SIMD):
// K independent accesses per loop iteration, results consumed once per iteration.  Patterns: every lane its own 8-byte word at a
// 66-double stride (the stage slots of csrc/small_kernel.hpp), 16 lanes x 4 sharing an address (matrix-layout broadcasts), all lanes one address.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/lds_issue_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int IT = 200, K = 8;
typedef double d2 __attribute__((ext_vector_type(2)));
template <int V>
__global__ void __launch_bounds__(64, 1) probe(unsigned long long *out, double *sink) {
    __shared__ __attribute__((aligned(16))) double lds[64 * 66 + 64];
    const int lane = threadIdx.x;
    for (int i = 0; i < 66; ++i) lds[lane * 66 + i] = 0.001 * ((lane + i) % 7);
    __syncthreads();
    const int pat = V % 3;   // 0 own slot, 1 quad-shared (lane >> 2), 2 one address
    double *base = lds + (pat == 0 ? lane * 66 : (pat == 1 ? (lane >> 2) * 66 : 0));
    double acc = 0.0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < IT; ++it) {
        if (V / 3 == 0) {   // ds_read_b64
            double v[K];
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = base[2 * j + (it & 1)];
#pragma unroll
            for (int j = 0; j < K; ++j) acc += v[j];
        } else if (V / 3 == 1) {   // ds_read_b128
            d2 v[K];
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = *(const d2 *)(base + 2 * j + 2 * (it & 1));
#pragma unroll
            for (int j = 0; j < K; ++j) acc += v[j].x + v[j].y;
        } else if (V / 3 == 2) {   // ds_write_b64
#pragma unroll
            for (int j = 0; j < K; ++j) base[2 * j + (it & 1)] = acc + j;
            acc += 1.0;
        } else {   // ds_write_b128
#pragma unroll
            for (int j = 0; j < K; ++j) {
                d2 w;
                w.x = acc, w.y = acc + j;
                *(d2 *)(base + 2 * j + 2 * (it & 1)) = w;
            }
            acc += 1.0;
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[V] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc + lds[lane];
}
int main() {
    unsigned long long *out, h[12];
    double *sink;
    hipMalloc(&out, 128);
    hipMalloc(&sink, 1024 * 64 * 8);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<3>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<4>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<5>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<6>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<7>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<8>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<9>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<10>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipLaunchKernelGGL(probe<11>, dim3(1024), dim3(64), 0, 0, out, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *op[4] = {"ds_read_b64", "ds_read_b128", "ds_write_b64", "ds_write_b128"}, *pt[3] = {"own slot (stride 66 doubles)", "4 lanes per address", "one address"};
    for (int i = 0; i < 12; ++i) printf("%-14s %-30s %6.1f cycles per instruction (incl. %d-op loop overhead)\n", op[i / 3], pt[i % 3], (double)h[i] / IT / K, K);
    return 0;
}
