// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane layout and cost.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/mfma_f64_16x16x4_probe.hip -o /tmp/probe16 && /tmp/probe16
// A one-hot A operand (lane la) and a one-hot B operand (lane lb) make exactly the D entries with A(i,k) * B(k,j) != 0 light up:
// from which (register, lane) pairs do, the maps lane -> (i, k) of A, lane -> (k, j) of B and (register, lane) -> (i, j) of D follow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            d4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r) out[((size_t)(la * 64 + lb) * 4 + r) * 64 + lane] = c[r];
        }
}
__global__ void timing(double *out, int n, unsigned long long *ticks) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    d4 c = {0, 0, 0, 0}, c2 = c, c3 = c, c4 = c;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    unsigned long long t1 = clock64();
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
    }
    unsigned long long t2 = clock64();
    out[threadIdx.x] = c[0] + c2[1] + c3[2] + c4[3];
    if (threadIdx.x == 0) ticks[0] = t1 - t0, ticks[1] = t2 - t1;
}
int main() {
    double *d; hipMalloc(&d, (size_t)64 * 64 * 4 * 64 * 8);
    probe<<<1, 64>>>(d);
    std::vector<double> h((size_t)64 * 64 * 4 * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // A lane la and B lane lb share a k iff some D entry is non-zero; the D entry tells (i of la, j of lb)
    int ai[64], ak[64], bk[64], bj[64];
    for (int l = 0; l < 64; ++l) ai[l] = ak[l] = bk[l] = bj[l] = -1;
    // hypothesis check: A(i = l % 16, k = l / 16), B(k = l / 16, j = l % 16), D[r](i = 4 * (l / 16) + r ... or r * 4 + l / 16, j = l % 16)
    int okA = 1, hyp1 = 1, hyp2 = 1;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            int cnt = 0, rr = -1, ll = -1;
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l)
                    if (h[((size_t)(la * 64 + lb) * 4 + r) * 64 + l] != 0.0) ++cnt, rr = r, ll = l;
            const bool same_k = la / 16 == lb / 16;
            if ((cnt == 1) != same_k) okA = 0;
            if (cnt == 1) {
                const int i = la % 16, j = lb % 16;
                if (!(ll % 16 == j && 4 * (ll / 16) + rr == i)) hyp1 = 0;
                if (!(ll % 16 == j && 4 * rr + ll / 16 == i)) hyp2 = 0;
                if (la < 2 && lb < 34 && (lb % 16) < 2) printf("A lane %d, B lane %d -> D reg %d lane %d\n", la, lb, rr, ll);
            }
        }
    printf("A(i = l %% 16, k = l / 16) and B(k = l / 16, j = l %% 16): %s\n", okA ? "yes" : "NO");
    printf("D[r] lane l = D(i = 4 (l / 16) + r, j = l %% 16): %s\n", hyp1 ? "yes" : "no");
    printf("D[r] lane l = D(i = 4 r + l / 16, j = l %% 16): %s\n", hyp2 ? "yes" : "no");
    for (int la : {0, 1, 16, 17, 35}) {
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)(la * 64 + (la / 16) * 16 + 3) * 4 + r) * 64 + l] != 0.0) printf("A lane %d x B lane %d -> D reg %d lane %d\n", la, (la / 16) * 16 + 3, r, l);
    }
    unsigned long long *t; hipMalloc(&t, 64);
    const int n = 10000;
    timing<<<1, 64>>>(d, n, t);
    unsigned long long ht[2]; hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
    printf("cycles per v_mfma_f64_16x16x4: dependent through C %.1f, 4 independent accumulators %.1f each\n", (double)ht[0] / n, (double)ht[1] / n / 4);
    return 0;
}
