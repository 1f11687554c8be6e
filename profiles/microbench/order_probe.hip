// order_probe.hip — where the packing-order kernel's ~15 us go (4096 keys, one workgroup): stages of the kernel switched on one by one,
// timed launch to launch in a stream.  Stage by stage (a copy of the kernel with early exits, round 3): launch floor 2.6 us, spread of
// the four coordinates 7.0, bucket atomics 1.0, crowded test 0.2, scan + scatter 1.4, per-bucket insertion sort 1.6, write-out 0.3.   hipcc --offload-arch=gfx950 -O3 -o order_probe order_probe.hip && ./order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mpc4rl_amd/csrc/order_kernel.hpp"
using namespace mpcrl;

__global__ void empty_kernel(int *p) { if (threadIdx.x == 9999) p[0] = 1; }

// variant: 4096 threads in 4 workgroups of 1024?  no — one workgroup, but ranks by all-pairs counting in LDS (no atomics, no scan, no sort)
__global__ void __launch_bounds__(1024) rank_kernel(const double *x0, int B, int nx, int dim, int *perm) {
    __shared__ double key[ORDER_MAX];
    const int tid = threadIdx.x;
    for (int i = tid; i < B; i += 1024) key[i] = x0[(size_t)i * nx + dim];
    __syncthreads();
    for (int i = tid; i < B; i += 1024) {
        const double k = key[i];
        int r = 0;
        for (int j = 0; j < B; ++j) r += (key[j] < k || (key[j] == k && j < i)) ? 1 : 0;
        perm[r] = i;
    }
}

int main() {
    const int B = 4096, nx = 4;
    std::vector<double> h(B * nx);
    for (int i = 0; i < B * nx; ++i) h[i] = (double)((i * 2654435761u) % 10007) / 10007.0;
    double *x; int *perm;
    hipMalloc(&x, B * nx * sizeof(double)); hipMalloc(&perm, B * sizeof(int));
    hipMemcpy(x, h.data(), B * nx * sizeof(double), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char *name, auto f) {
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(a);
        for (int i = 0; i < 500; ++i) f();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %.2f us per launch\n", name, ms / 500 * 1e3);
    };
    time("empty kernel, 1 x 64", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, perm); });
    time("empty kernel, 1 x 1024", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, 0, perm); });
    double *state; hipMalloc(&state, 4 * sizeof(double));
    time("order_kernel, from scratch (refresh)", [&] { hipLaunchKernelGGL(order_kernel, dim3(1), dim3(ORDER_NT), 0, 0, x, B, nx, perm, state, 1); });
    time("order_kernel, remembered coordinate", [&] { hipLaunchKernelGGL(order_kernel, dim3(1), dim3(ORDER_NT), 0, 0, x, B, nx, perm, state, 0); });
    return 0;
}
