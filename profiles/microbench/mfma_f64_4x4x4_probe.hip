// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout and issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 16; ++la)
        for (int lb = 0; lb < 16; ++lb) {
            const double a = (lane % 16 == la) ? 1.0 + lane / 16 : 0.0;   // block b carries value 1 + b
            const double b = (lane % 16 == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 16 + lb) * 64 + lane] = d;
        }
}
__global__ void timing(double *out, int n, unsigned long long *ticks) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3, c = 0.0, c2 = 0.0, c3 = 0.0, c4 = 0.0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);   // dependent through C
    unsigned long long t1 = clock64();
    for (int i = 0; i < n; ++i) {   // 4 independent accumulators
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    }
    unsigned long long t2 = clock64();
    double x = a;
    for (int i = 0; i < n; ++i) x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b, c, 0, 0, 0);   // dependent through A (result -> operand)
    unsigned long long t3 = clock64();
    double f = a;
    for (int i = 0; i < n; ++i) f = fma(f, b, c);   // dependent VALU fma chain
    unsigned long long t4 = clock64();
    out[threadIdx.x] = c + c2 + c3 + c4 + x + f;
    if (threadIdx.x == 0) ticks[0] = t1 - t0, ticks[1] = t2 - t1, ticks[2] = t3 - t2, ticks[3] = t4 - t3;
}
int main() {
    double *d; hipMalloc(&d, 256 * 64 * 8);
    probe<<<1, 64>>>(d);
    std::vector<double> h(256 * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // for each A-lane la and B-lane lb (within block 0): which D lanes are non-zero
    printf("la lb -> D lanes (block0 lanes 0..15), value\n");
    for (int la = 0; la < 16; ++la)
        for (int lb = 0; lb < 16; ++lb) {
            bool any = false;
            for (int l = 0; l < 64; ++l)
                if (h[(la * 16 + lb) * 64 + l] != 0.0) {
                    if (!any) printf("%2d %2d :", la, lb);
                    any = true;
                    printf(" %d(%.0f)", l, h[(la * 16 + lb) * 64 + l]);
                }
            if (any) printf("\n");
        }
    unsigned long long *t; hipMalloc(&t, 64);
    const int n = 10000;
    timing<<<1, 64>>>(d, n, t);
    unsigned long long ht[4]; hipMemcpy(ht, t, 32, hipMemcpyDeviceToHost);
    printf("cycles per mfma: dependent-C %.1f, 4 independent %.1f each, dependent-A %.1f; dependent v_fma_f64 %.1f\n", (double)ht[0] / n, (double)ht[1] / n / 4, (double)ht[2] / n, (double)ht[3] / n);
    return 0;
}
