// mx_loop_probe — cycles of the matrix-layout sweeps of the cartpole solver (csrc/small_kernel.hpp: mx_factor, mx_chain) in isolation:
// one wavefront per workgroup, the slots filled with a plausible linearisation, each sweep repeated REPS times between two
// s_memtime reads.  Shows what a sweep costs when the rest of the solver does not compete for registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form profiles/microbench/mx_loop_probe.hip -o /tmp/mx_loop_probe && /tmp/mx_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mpc4rl_amd/csrc/small_kernel.hpp"
using namespace mpcrl;
using S_t = SmallSolver<CartpoleDev>;
constexpr int REPS = 200;

__global__ void __launch_bounds__(64, 1) probe(SmallSpec sp, unsigned long long *out, double *sink) {
    __shared__ __attribute__((aligned(16))) double lds[S_t::MX_LDS];
    const int lane = threadIdx.x, lpi = sp.N + 1, slot = lane / lpi, k = lane - slot * lpi;
    S_t S(sp, k, lpi, slot * lpi);
    S.ms = lds;
    S.qmode = false;
    // a stable closed loop: A = I + 0.1 * pattern, B = 0.1, Hxx + D = 2 I, Huu + D = 1, small right-hand sides
    double *sl = lds + lane * S_t::MSLOT;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            sl[2 * (4 * i + j)] = (i == j ? 1.0 : 0.0) + 0.01 * ((i + 2 * j + k) % 5);
            sl[2 * (4 * i + j) + 1] = i == j ? 2.0 : 0.05;
        }
    for (int i = 0; i < 4; ++i) {
        sl[32 + 4 * i] = 0.1 + 0.01 * i, sl[33 + 4 * i] = 0.02, sl[34 + 4 * i] = 0.01 * (i + 1), sl[35 + 4 * i] = 0.03 * (k % 3);
        sl[48 + 2 * i] = i == 0 ? 1.5 : (i == 1 ? 0.2 : 0.0), sl[49 + 2 * i] = 0.02;
    }
    for (int i = 56; i < 66; ++i) sl[i] = 0.0;
    S_t::wave_lds_sync();
    unsigned long long t0, t1, t2, t3;
    t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
        S.template mx_factor<false>();
        S_t::wave_lds_sync();
    }
    t1 = clock64();
    for (int r = 0; r < REPS; ++r) {
        S.template mx_chain<true>();
        S_t::wave_lds_sync();
    }
    t2 = clock64();
    for (int r = 0; r < REPS; ++r) {
        S.template mx_chain<false>();
        S_t::wave_lds_sync();
    }
    t3 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0, out[1] = t2 - t1, out[2] = t3 - t2;
    sink[blockIdx.x * 64 + lane] = sl[57] + sl[49] + sl[33];
}

int main(int argc, char **argv) {
    SmallSpec sp{};
    sp.N = 20;
    unsigned long long *out;
    double *sink;
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;   // 1024 = one wavefront on every SIMD, as in the solver; 256 = one per CU (the LDS to itself)
    hipMalloc(&out, 64);
    hipMalloc(&sink, blocks * 64 * sizeof(double));
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, sp, out, sink);
        hipDeviceSynchronize();
    }
    unsigned long long h[3];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    std::vector<double> hs(64);
    hipMemcpy(hs.data(), sink, 64 * sizeof(double), hipMemcpyDeviceToHost);
    const char *names[3] = {"factor sweep", "forward chain", "backward chain"};
    for (int i = 0; i < 3; ++i)
        printf("%-15s %8.1f ticks per sweep  %6.1f per stage step (N = %d)\n", names[i], (double)h[i] / REPS, (double)h[i] / REPS / sp.N, sp.N);
    printf("%d workgroups of one wavefront\n", blocks);
    printf("(sink %g %g; s_memtime ticks: 100 MHz constant clock on gfx9 unless the clock64 source is the shader clock)\n", hs[0], hs[21]);
    return 0;
}
