// microbenchmark: cost of fp64 FMA chains under different EXEC masks (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters) {
    const int lane = threadIdx.x & 63;
    bool act;
    if (MODE == 0) act = true;
    else if (MODE == 1) act = lane < 3;
    else if (MODE == 2) act = (lane == 0 || lane == 21 || lane == 42);
    else if (MODE == 3) act = lane == 0;
    else if (MODE == 4) act = lane < 16;
    else if (MODE == 5) act = lane < 8;
    else if (MODE == 6) act = lane < 4;
    else if (MODE == 7) act = (lane & 15) == 0;
    else if (MODE == 8) act = lane < 32;
    else if (MODE == 9) act = lane < 12;
    else act = (lane & 3) == 0;
    double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    if (act) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
                a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char *name, double *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 1024;
    k<MODE><<<blocks, 64>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fmas = (double)iters * 128;
    printf("%-28s %8.3f ms   %.2f ns per wave-FMA-instruction\n", name, ms, ms * 1e6 / fmas);
}
int main() {
    double *d; hipMalloc(&d, 1024 * 64 * 8);
    run<0>("all 64 lanes", d);
    run<4>("lanes 0..15", d);
    run<1>("lanes 0,1,2 (one quarter)", d);
    run<2>("lanes 0,21,42 (3 quarters)", d);
    run<3>("lane 0", d);
    run<5>("lanes 0..7", d);
    run<9>("lanes 0..11", d);
    run<6>("lanes 0..3", d);
    run<7>("lanes 0,16,32,48", d);
    run<8>("lanes 0..31", d);
    run<10>("every 4th lane (16 lanes)", d);
    return 0;
}
