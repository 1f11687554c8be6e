// mfma_chain_probe — what a step of a dependent v_mfma_f64_4x4x4 chain costs on a lone wavefront (gfx950), by ingredient:
//   0 bare chain V = mfma(X, V, C)                         1 + ds_write of V every step
//   2 + two ds_reads per step (operands 3 steps ahead)     3 = 2 + the write        4 = 3 with the write one step late
//   5 dependent fp64 fma chain (latency of one VALU op)    6 dependent v_rcp_f64    7 dependent DPP quad_perm mov + fma
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form profiles/microbench/mfma_chain_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int STEPS = 240;
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double qb0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int V>
__global__ void __launch_bounds__(64, 1) probe(unsigned long long *out, double *sink) {
    __shared__ double lds[64 * 66];
    const int lane = threadIdx.x;
    for (int i = 0; i < 66; ++i) lds[lane * 66 + i] = 0.001 * ((lane + i) % 7);
    __syncthreads();
    double v = 0.5 + 0.001 * lane, x = 0.01, c = 0.001;
    const double *rd = lds + (lane & 15) * 66;
    double *wr = lds + lane * 66 + 60;
    double x0 = rd[0], c0 = rd[1], x1 = rd[66], c1 = rd[67], x2 = rd[132], c2 = rd[133], prev = 0.0;
    const unsigned long long t0 = clock64();
    if (V <= 4) {
        for (int s = 0; s < STEPS; s += 3) {
            const double *nx = rd + ((s + 3) % 48) * 66;
            if (V == 0) { v = mfma4(x, v, c); v = mfma4(x, v, c); v = mfma4(x, v, c); }
            if (V == 1) { v = mfma4(x, v, c); wr[0] = v; v = mfma4(x, v, c); wr[1] = v; v = mfma4(x, v, c); wr[2] = v; }
            if (V == 2) { v = mfma4(x0, v, c0); x0 = nx[0], c0 = nx[1]; v = mfma4(x1, v, c1); x1 = nx[66], c1 = nx[67]; v = mfma4(x2, v, c2); x2 = nx[132], c2 = nx[133]; }
            if (V == 3) { v = mfma4(x0, v, c0); wr[0] = v; x0 = nx[0], c0 = nx[1]; v = mfma4(x1, v, c1); wr[1] = v; x1 = nx[66], c1 = nx[67]; v = mfma4(x2, v, c2); wr[2] = v; x2 = nx[132], c2 = nx[133]; }
            if (V == 4) {
                double n = mfma4(x0, v, c0); wr[0] = v; v = n; x0 = nx[0], c0 = nx[1];
                n = mfma4(x1, v, c1); wr[1] = v; v = n; x1 = nx[66], c1 = nx[67];
                n = mfma4(x2, v, c2); wr[2] = v; v = n; x2 = nx[132], c2 = nx[133];
            }
        }
    } else {
        for (int s = 0; s < STEPS; ++s) {
            if (V == 5) v = fma(v, x, c);
            if (V == 6) v = __builtin_amdgcn_rcp(v) + c;
            if (V == 7) v = fma(qb0(v), x, c);
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[V] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = v + prev + lds[lane * 66 + 61];
}
int main() {
    unsigned long long *out, h[8];
    double *sink;
    hipMalloc(&out, 64);
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
        hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[8] = {"bare MFMA chain", "+ ds_write", "+ 2 ds_read (3 ahead)", "+ reads + write", "+ reads + late write", "fp64 fma chain", "v_rcp_f64 + add chain", "DPP mov + fma chain"};
    for (int i = 0; i < 8; ++i) printf("%d %-24s %7.1f cycles per step\n", i, nm[i], (double)h[i] / STEPS);
    return 0;
}
