// The critic side of one TD3 update in two launches (round 6): what stable_baselines3's TD3.train does between the target actor's action
// and the critic optimiser's step (the reference pulls stable_baselines3 in unpinned, pyproject.toml:10; its critics are the Q(s, a) MLPs of
// rlmpc/td3/policies.py:47-122: n_critics networks [obs | action] -> 64 -> 64 -> 1 with ReLU) —
//     ok_b   = ok_u[b] and every entry of the stored transition is finite            (a failed target solve or a poisoned row is SELECTED out)
//     q'_c   = Q'_c(s'_b, a'_b)   for the target critics,      y_b = r_b + gamma (1 - done_b) min_c q'_c
//     e_cb   = ok_b ? Q_c(s_b, a_b) - y_b : 0,      loss = sum_c sum_b e_cb^2 / max(1, sum_b ok_b)
//     grad   = d loss / d (parameters of Q)                                            (the backward pass of the two MLPs by hand)
// — ~60 framework launches (rocBLAS products of [4096 x 64] by [64 x 64], ReLUs, their masks, bias reductions, the loss pieces, the
// concatenation of 12 gradient tensors: 361 us per update, graph-replayed) as critic_td_partial_kernel + critic_td_reduce_kernel.
//
// critic_td_partial_kernel: one workgroup per CRITIC_S = 16 transitions, one wavefront per critic, lane j = hidden unit j.  A lane keeps row j of
// the 64 x 64 weight in registers for the forward passes (target, then online) and column j for the backward pass; activations of the
// workgroup's transitions live in LDS and are read as broadcasts.  fp32 FMAs on the vector ALU: the products are 16 x 64 x 64 per
// workgroup — below one matrix-core tile's worth of latency to set up, and 4096 transitions are 256 workgroups, one per CU.  Every
// workgroup writes its partial gradient; critic_td_reduce_kernel sums them in a fixed order (fp64 accumulation, no atomics: the result
// does not depend on scheduling), divides by the count and writes the fp64 message the loop all-reduces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcrl {

constexpr int CRITIC_H = 64;      // hidden width (both layers): the lane count of a wavefront
constexpr int CRITIC_S = 16;      // transitions per workgroup
constexpr int CRITIC_DMAX = 64;   // inputs of the first layer (nx + nu)

struct CriticArgs {
    const float *rows;        // [B][row_stride]: obs (nx) | next obs (nx) | action (nu) | reward | done | ...
    int row_stride, row_len;  // row_len = 2 nx + nu + 2: the entries tested for finiteness
    int B, nx, nu, n_critics;
    const float *a_next;      // [B][nu]
    const uint8_t *ok_u;      // [B] or nullptr
    const float *params, *params_target;   // per critic: W1 [64][D] | b1 [64] | W2 [64][64] | b2 [64] | W3 [64] | b3 [1]
    float gamma;
    float *partial;           // [n_blocks][n_params + 2]   (per block: the gradient partial, then loss partial, then ok count)
    uint8_t *ok_out;          // [B] or nullptr
};

__device__ inline int critic_params_per_net(int D) { return CRITIC_H * D + CRITIC_H + CRITIC_H * CRITIC_H + CRITIC_H + CRITIC_H + 1; }

// sum over the 64 lanes, every lane gets it
__device__ inline float wave_sum64(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ void __launch_bounds__(128) critic_td_partial_kernel(const CriticArgs a) {
    constexpr int H = CRITIC_H, S = CRITIC_S;
    const int D = a.nx + a.nu, c = threadIdx.x >> 6, j = threadIdx.x & 63, b0 = blockIdx.x * S;
    const int npn = critic_params_per_net(D), n_params = npn * a.n_critics;
    __shared__ float xt[S][CRITIC_DMAX], xo[S][CRITIC_DMAX];      // inputs of the target pass (s', a') and of the online pass (s, a)
    __shared__ float rew[S], dn[S], ys[S], okf[S];
    __shared__ float qn[2][S];
    __shared__ __attribute__((aligned(16))) float h1s[2][S][H];    // first-layer activations, per critic
    __shared__ float ps[2][S][H + 1];                              // per-lane pieces of the output sum
    __shared__ __attribute__((aligned(16))) float g2s[2][S][H];    // the second layer's pre-activation gradient
    __shared__ float w1s[2][H][CRITIC_DMAX + 1];                   // first-layer weight of the pass at hand; then its gradient accumulator
    // ---- the workgroup's transitions
    if (threadIdx.x < S) {
        const int s = threadIdx.x, b = b0 + s;
        bool ok = b < a.B && (!a.ok_u || a.ok_u[b]);
        if (b < a.B) {
            const float *r = a.rows + (long)b * a.row_stride;
            for (int i = 0; i < a.row_len; ++i) ok = ok && isfinite(r[i]);
            for (int i = 0; i < a.nu; ++i) ok = ok && isfinite(a.a_next[(long)b * a.nu + i]);
            if (a.ok_out) a.ok_out[b] = ok ? 1 : 0;
        }
        okf[s] = ok ? 1.0f : 0.0f;
        const float *r = a.rows + (long)(b < a.B ? b : 0) * a.row_stride;
        rew[s] = ok ? r[2 * a.nx + a.nu] : 0.0f, dn[s] = ok ? r[2 * a.nx + a.nu + 1] : 0.0f;
        for (int d = 0; d < D; ++d) {
            xo[s][d] = !ok ? 0.0f : (d < a.nx ? r[d] : r[2 * a.nx + (d - a.nx)]);
            xt[s][d] = !ok ? 0.0f : (d < a.nx ? r[a.nx + d] : a.a_next[(long)b * a.nu + (d - a.nx)]);
        }
    }
    const bool live = c < a.n_critics;      // (n_critics = 1: the second wavefront only keeps the barriers company)
    float w2r[H], h2r[S];
    float b1 = 0.0f, b2 = 0.0f, w3 = 0.0f, b3 = 0.0f;
    auto load_net = [&](const float *p) {
        const float *W1 = p, *B1 = W1 + H * D, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H, *B3 = W3 + H;
        for (int d = 0; d < D; ++d) w1s[c][j][d] = W1[j * D + d];
        b1 = B1[j], b2 = B2[j], w3 = W3[j], b3 = B3[0];
#pragma unroll
        for (int i = 0; i < H; ++i) w2r[i] = W2[j * H + i];      // (a net is an odd number of floats: no vector alignment to rely on)
    };
    // forward pass of the net at hand over the S transitions: h1 to LDS, h2 to registers, the output's per-lane pieces to LDS
    auto forward = [&](const float (*x)[CRITIC_DMAX]) {
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            float z = b1;
            for (int d = 0; d < D; ++d) z = fmaf(w1s[c][j][d], x[s][d], z);
            h1s[c][s][j] = fmaxf(z, 0.0f);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float z = b2;
#pragma unroll
            for (int i = 0; i < H; i += 4) {
                const float4 h = *(const float4 *)&h1s[c][s][i];
                z = fmaf(w2r[i], h.x, z), z = fmaf(w2r[i + 1], h.y, z), z = fmaf(w2r[i + 2], h.z, z), z = fmaf(w2r[i + 3], h.w, z);
            }
            h2r[s] = fmaxf(z, 0.0f);
            ps[c][s][j] = w3 * h2r[s];
        }
        __syncthreads();
    };
    // q[s] for s = lane / 4: the four lanes of a quad sum a quarter of the pieces each
    auto output = [&]() {
        const int s = j >> 2, k = j & 3;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) q += ps[c][s][k * 16 + i];
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        return q + b3;
    };
    __syncthreads();
    // ---- target critics: y
    if (live) load_net(a.params_target + (long)c * npn);
    forward(xt);
    {
        const float q = output();
        if (live && (j & 3) == 0) qn[c][j >> 2] = q;
    }
    __syncthreads();
    if (threadIdx.x < S) {
        const int s = threadIdx.x;
        const float qm = a.n_critics > 1 ? fminf(qn[0][s], qn[1][s]) : qn[0][s];
        ys[s] = okf[s] != 0.0f ? rew[s] + a.gamma * (1.0f - dn[s]) * qm : 0.0f;
    }
    __syncthreads();
    // ---- online critics: error, loss
    if (live) load_net(a.params + (long)c * npn);
    forward(xo);
    float loss = 0.0f;
    {
        const int s = j >> 2;
        const float q = output();
        const float e = okf[s] != 0.0f ? q - ys[s] : 0.0f;
        if ((j & 3) == 0) qn[c][s] = 2.0f * e, loss = e * e;      // dq of the unscaled loss (the count divides in the reduction)
    }
    loss = wave_sum64(loss);
    __syncthreads();
    // ---- backward
    // second layer's pre-activation gradient g2[s][j] to LDS, the output layer's gradients on the way
    float dw3 = 0.0f, db3 = 0.0f, db2 = 0.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float dq = qn[c][s];
        const float g2 = h2r[s] > 0.0f ? dq * w3 : 0.0f;
        dw3 = fmaf(dq, h2r[s], dw3), db3 += dq, db2 += g2;
        g2s[c][s][j] = g2;
    }
    // column j of W2 (coalesced over the lanes), and the first layer's gradient accumulator in place of its weight
    float w2c[H];
    {
        const float *W2 = a.params + (long)(live ? c : 0) * npn + H * D + H;
#pragma unroll
        for (int i = 0; i < H; ++i) w2c[i] = W2[i * H + j];
    }
    for (int d = 0; d < D; ++d) w1s[c][j][d] = 0.0f;
    __syncthreads();
    float dw2[H];
#pragma unroll
    for (int i = 0; i < H; ++i) dw2[i] = 0.0f;
    float db1 = 0.0f;
#pragma unroll 2
    for (int s = 0; s < S; ++s) {
        const float g2 = g2s[c][s][j];
        float g1 = 0.0f;
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            const float4 h = *(const float4 *)&h1s[c][s][i], g = *(const float4 *)&g2s[c][s][i];
            dw2[i] = fmaf(g2, h.x, dw2[i]), dw2[i + 1] = fmaf(g2, h.y, dw2[i + 1]), dw2[i + 2] = fmaf(g2, h.z, dw2[i + 2]), dw2[i + 3] = fmaf(g2, h.w, dw2[i + 3]);
            g1 = fmaf(w2c[i], g.x, g1), g1 = fmaf(w2c[i + 1], g.y, g1), g1 = fmaf(w2c[i + 2], g.z, g1), g1 = fmaf(w2c[i + 3], g.w, g1);
        }
        g1 = h1s[c][s][j] > 0.0f ? g1 : 0.0f;
        db1 += g1;
        for (int d = 0; d < D; ++d) w1s[c][j][d] = fmaf(g1, xo[s][d], w1s[c][j][d]);
    }
    // ---- the workgroup's partial
    float *out = a.partial + (long)blockIdx.x * (n_params + 2);
    if (live) {
        float *o = out + (long)c * npn;
        for (int d = 0; d < D; ++d) o[j * D + d] = w1s[c][j][d];
        o += H * D;
        o[j] = db1;
        o += H;
#pragma unroll
        for (int i = 0; i < H; ++i) o[j * H + i] = dw2[i];
        o += H * H;
        o[j] = db2;
        o += H;
        o[j] = dw3;
        o += H;
        if (j == 0) o[0] = db3;
    }
    // loss partial (both critics), ok count
    if (j == 0) qn[c][0] = live ? loss : 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[n_params] = qn[0][0] + qn[1][0];
        float n = 0.0f;
        for (int s = 0; s < S; ++s) n += okf[s];
        out[n_params + 1] = n;
    }
}

// grad[t] = out_scale / max(1, n_ok) * sum_blocks partial[block][t]  (fp64), loss = sum / max(1, n_ok)
__global__ void __launch_bounds__(256) critic_td_reduce_kernel(const float *partial, int n_blocks, int n_params, double out_scale, double *grad, float *loss_out) {
    __shared__ double red[256];
    const int stride = n_params + 2;
    double n = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) n += partial[(long)b * stride + n_params + 1];
    red[threadIdx.x] = n;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    const double cnt = red[0] > 1.0 ? red[0] : 1.0;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t <= n_params) {
        double acc = 0.0;
        for (int b = 0; b < n_blocks; ++b) acc += partial[(long)b * stride + t];
        if (t < n_params)
            grad[t] = acc * out_scale / cnt;
        else if (loss_out)
            *loss_out = (float)(acc / cnt);
    }
}

// dQ_1/da at (s_b, a_b) for the deterministic policy gradient (what autograd of ContinuousCritic.q1_forward, rlmpc/td3/policies.py:68-76,
// gives): forward of the first critic, then back through the two ReLU layers to the action inputs.  Rows with ok[b] = 0 get 0 (their
// inputs are read as 0, like the torch expression that selects them out).  One workgroup of one wavefront per 16 rows.
struct CriticDqdaArgs {
    const float *obs;     // [B][obs_stride]: the first nx entries
    int obs_stride, B, nx, nu;
    const float *act;     // [B][nu]
    const uint8_t *ok;    // [B] or nullptr
    const float *params;  // the first critic
    float *dq_da;         // [B][nu]
    uint8_t *ok_out;      // [B] or nullptr
};

__global__ void __launch_bounds__(64) critic_dqda_kernel(const CriticDqdaArgs a) {
    constexpr int H = CRITIC_H, S = CRITIC_S;
    const int D = a.nx + a.nu, j = threadIdx.x, b0 = blockIdx.x * S;
    __shared__ float x[S][CRITIC_DMAX], okf[S];
    __shared__ __attribute__((aligned(16))) float h1s[S][H], g2s[S][H];
    __shared__ float g1s[S][H + 1];
    if (j < S) {
        const int b = b0 + j;
        bool ok = b < a.B && (!a.ok || a.ok[b]);
        if (ok) {
            for (int d = 0; d < a.nx; ++d) ok = ok && isfinite(a.obs[(long)b * a.obs_stride + d]);
            for (int d = 0; d < a.nu; ++d) ok = ok && isfinite(a.act[(long)b * a.nu + d]);
        }
        okf[j] = ok ? 1.0f : 0.0f;
        if (a.ok_out && b < a.B) a.ok_out[b] = ok ? 1 : 0;
        for (int d = 0; d < D; ++d) x[j][d] = !ok ? 0.0f : (d < a.nx ? a.obs[(long)b * a.obs_stride + d] : a.act[(long)b * a.nu + (d - a.nx)]);
    }
    const float *W1 = a.params, *B1 = W1 + H * D, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H;
    float w2r[H], w2c[H], h2r[S];
#pragma unroll
    for (int i = 0; i < H; ++i) w2r[i] = W2[j * H + i], w2c[i] = W2[i * H + j];
    const float b1 = B1[j], b2 = B2[j], w3 = W3[j];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        float z = b1;
        for (int d = 0; d < D; ++d) z = fmaf(W1[j * D + d], x[s][d], z);
        h1s[s][j] = fmaxf(z, 0.0f);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float z = b2;
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            const float4 h = *(const float4 *)&h1s[s][i];
            z = fmaf(w2r[i], h.x, z), z = fmaf(w2r[i + 1], h.y, z), z = fmaf(w2r[i + 2], h.z, z), z = fmaf(w2r[i + 3], h.w, z);
        }
        g2s[s][j] = z > 0.0f ? w3 : 0.0f;            // d q / d z2_j
    }
    __syncthreads();
#pragma unroll 2
    for (int s = 0; s < S; ++s) {
        float g1 = 0.0f;
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            const float4 g = *(const float4 *)&g2s[s][i];
            g1 = fmaf(w2c[i], g.x, g1), g1 = fmaf(w2c[i + 1], g.y, g1), g1 = fmaf(w2c[i + 2], g.z, g1), g1 = fmaf(w2c[i + 3], g.w, g1);
        }
        g1s[s][j] = h1s[s][j] > 0.0f ? g1 : 0.0f;    // d q / d z1_j
    }
    __syncthreads();
    // dq/da_u = sum_i W1[i][nx + u] g1_i: lane = (row s, quarter k)
    const int s = j >> 2, k = j & 3, b = b0 + s;
    for (int u = 0; u < a.nu; ++u) {
        float v = 0.0f;
        for (int i = 0; i < 16; ++i) v = fmaf(W1[(k * 16 + i) * D + a.nx + u], g1s[s][k * 16 + i], v);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (k == 0 && b < a.B) a.dq_da[(long)b * a.nu + u] = okf[s] != 0.0f ? v : 0.0f;
    }
}

}  // namespace mpcrl
