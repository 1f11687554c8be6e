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
// critic_td_partial_kernel: one workgroup per CRITIC_S = 16 transitions, lane j = hidden unit j; per critic one wavefront, or two that share
// its transitions (nx + nu <= 16: four wavefronts, one per SIMD).  A lane keeps row j of the 64 x 64 weight in registers for the forward
// passes (target, then online) and column j for the backward pass — both come from one coalesced read of the matrix through LDS —
// and the first-layer weights and their gradients too (up to 16 inputs); activations of the workgroup's transitions live in LDS and are
// read as broadcasts.  fp32 FMAs on the vector ALU: the kernel is bound by the issue rate of a lone wavefront per SIMD (phase clocks:
// forward 10 k cycles per net, backward 26 k with two wavefronts; 45 us -> 27 us with four), not by bytes or flops — 4096 transitions
// are 256 workgroups, one per CU.  Every workgroup writes its partial gradient (the 64 x 64 block transposed: coalesced);
// critic_td_reduce_kernel sums them in a fixed order (fp64 accumulation, no atomics: the result does not depend on scheduling),
// divides by the count and writes the fp64 message the loop all-reduces.
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

// HV = 2 (nx + nu <= 16): FOUR wavefronts per workgroup — each critic's 16 transitions are split between two of them (all four SIMDs of
// the CU issue; the halves' gradients are added through LDS before the partial is written, the target and the online weights are
// fetched by one half each).  HV = 1: two wavefronts, any nx + nu <= 64.
template <int HV>
__global__ void __launch_bounds__(128 * HV) critic_td_partial_kernel(const CriticArgs a) {
    constexpr int H = CRITIC_H, S = CRITIC_S, RMAX = 2 * CRITIC_DMAX + 2, NT = 128 * HV, SH = S / HV;
    constexpr int DR = 16;      // first-layer inputs whose weights (and weight gradients) a lane keeps in registers; the rest goes through LDS
    const int D = a.nx + a.nu, w = threadIdx.x >> 6, c = w & 1, half = w >> 1, j = threadIdx.x & 63, b0 = blockIdx.x * S;
    const int s0 = half * SH, s1 = s0 + SH;      // this wavefront's transitions
    const int npn = critic_params_per_net(D), n_params = npn * a.n_critics;
    __shared__ float rowb[S][RMAX], anb[S][CRITIC_DMAX];          // the workgroup's transitions as stored, the target actions
    __shared__ __attribute__((aligned(16))) float xt[S][CRITIC_DMAX], xo[S][CRITIC_DMAX];      // inputs of the target pass (s', a') and of the online pass (s, a)
    __shared__ float rew[S], dn[S], ys[S], okf[S];
    __shared__ float qn[2][S], lossw[4];
    __shared__ __attribute__((aligned(16))) float h1s[2][S][H];    // first-layer activations, per critic
    __shared__ float ps[2][S][H + 1];                              // per-lane pieces of the output sum
    __shared__ __attribute__((aligned(16))) float g2s[2][S][H];    // second-layer activations (forward), then the second layer's pre-activation gradient
    __shared__ float w1s[2][H][CRITIC_DMAX + 1];                   // first-layer weight of the pass at hand; then its gradient accumulator
    __shared__ float w2s[2][2][H][H + 1];                          // second-layer weights, [target / online][critic] (read by rows and by columns)
    // ---- the workgroup's transitions: every load in flight at once (128 lanes), then one lane per transition looks at them in LDS
    for (int e = threadIdx.x; e < S * a.row_len; e += NT) {
        const int s = e / a.row_len, i = e - s * a.row_len, b = b0 + s;
        rowb[s][i] = b < a.B ? a.rows[(long)b * a.row_stride + i] : 0.0f;
    }
    for (int e = threadIdx.x; e < S * a.nu; e += NT) {
        const int s = e / a.nu, i = e - s * a.nu, b = b0 + s;
        anb[s][i] = b < a.B ? a.a_next[(long)b * a.nu + i] : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < S) {
        const int s = threadIdx.x, b = b0 + s;
        bool ok = b < a.B && (!a.ok_u || a.ok_u[b]);
        for (int i = 0; i < a.row_len; ++i) ok = ok & (bool)isfinite(rowb[s][i]);
        for (int i = 0; i < a.nu; ++i) ok = ok & (bool)isfinite(anb[s][i]);
        if (a.ok_out && b < a.B) a.ok_out[b] = ok ? 1 : 0;
        okf[s] = ok ? 1.0f : 0.0f;
        rew[s] = ok ? rowb[s][2 * a.nx + a.nu] : 0.0f, dn[s] = ok ? rowb[s][2 * a.nx + a.nu + 1] : 0.0f;
        for (int d = 0; d < D; ++d) {
            xo[s][d] = !ok ? 0.0f : (d < a.nx ? rowb[s][d] : rowb[s][2 * a.nx + (d - a.nx)]);
            xt[s][d] = !ok ? 0.0f : (d < a.nx ? rowb[s][a.nx + d] : anb[s][d - a.nx]);
        }
        for (int d = D; d < DR; ++d) xo[s][d] = xt[s][d] = 0.0f;
    }
    const bool live = c < a.n_critics;      // (n_critics = 1: the second wavefront only keeps the barriers company)
    float w2r[H];
    float b1 = 0.0f, b2 = 0.0f, w3 = 0.0f, b3 = 0.0f;
    // Both nets' weights are asked for at once, up front (W2 in coalesced rows: lane j gets W2[i][j], through LDS, back as row j in
    // registers when its pass begins); the first DR columns of W1 live in registers, the rest in LDS
    float w1r[DR];
    if (live) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (HV == 2 && k != half) continue;
            const float *W2 = (k ? a.params : a.params_target) + (long)c * npn + H * D + H;
#pragma unroll
            for (int i = 0; i < H; ++i) w2s[k][c][i][j] = W2[i * H + j];
        }
    }
    auto load_net = [&](int k) {
        const float *p = (k ? a.params : a.params_target) + (long)(live ? c : 0) * npn;
        const float *W1 = p, *B1 = W1 + H * D, *B2 = B1 + H + H * H, *W3 = B2 + H, *B3 = W3 + H;
#pragma unroll
        for (int d = 0; d < DR; ++d) w1r[d] = d < D ? W1[j * D + d] : 0.0f;
        for (int d = DR; d < D; ++d) w1s[c][j][d] = W1[j * D + d];
        b1 = B1[j], b2 = B2[j], w3 = W3[j], b3 = B3[0];
#pragma unroll
        for (int i = 0; i < H; ++i) w2r[i] = w2s[k][c][j][i];
    };
    // forward pass of the net at hand over the S transitions: h1, h2 and the output's per-lane pieces to LDS
    auto forward = [&](const float (*x)[CRITIC_DMAX]) {
        // (the loops over the transitions stay rolled: straight-line code that runs once is paid in instruction fetches)
#pragma unroll 2
        for (int s = s0; s < s1; ++s) {
            float z = b1;
#pragma unroll
            for (int d = 0; d < DR; d += 4) {
                if (d >= D) break;      // (uniform)
                const float4 v = *(const float4 *)&x[s][d];
                z = fmaf(w1r[d], v.x, z), z = fmaf(w1r[d + 1], v.y, z), z = fmaf(w1r[d + 2], v.z, z), z = fmaf(w1r[d + 3], v.w, z);
            }
            for (int d = DR; d < D; ++d) z = fmaf(w1s[c][j][d], x[s][d], z);
            h1s[c][s][j] = fmaxf(z, 0.0f);
        }
        __syncthreads();
#pragma unroll 2
        for (int s = s0; s < s1; ++s) {
            float z0 = b2, z1 = 0.0f;
#pragma unroll
            for (int i = 0; i < H; i += 8) {
                const float4 h = *(const float4 *)&h1s[c][s][i], k = *(const float4 *)&h1s[c][s][i + 4];
                z0 = fmaf(w2r[i], h.x, z0), z0 = fmaf(w2r[i + 1], h.y, z0), z0 = fmaf(w2r[i + 2], h.z, z0), z0 = fmaf(w2r[i + 3], h.w, z0);
                z1 = fmaf(w2r[i + 4], k.x, z1), z1 = fmaf(w2r[i + 5], k.y, z1), z1 = fmaf(w2r[i + 6], k.z, z1), z1 = fmaf(w2r[i + 7], k.w, z1);
            }
            const float h2 = fmaxf(z0 + z1, 0.0f);
            g2s[c][s][j] = h2;
            ps[c][s][j] = w3 * h2;
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
    load_net(0);
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
    load_net(1);
    forward(xo);
    float loss = 0.0f;
    {
        const int s = j >> 2;
        const float q = output();
        const float e = okf[s] != 0.0f ? q - ys[s] : 0.0f;
        if ((j & 3) == 0) {
            qn[c][s] = 2.0f * e;      // dq of the unscaled loss (the count divides in the reduction); both halves write the same value
            if (s >= s0 && s < s1) loss = e * e;
        }
    }
    loss = wave_sum64(loss);
    __syncthreads();
    // ---- backward
    // second layer's pre-activation gradient g2[s][j] to LDS, the output layer's gradients on the way
    float dw3 = 0.0f, db3 = 0.0f, db2 = 0.0f;
#pragma unroll 1
    for (int s = s0; s < s1; ++s) {
        const float dq = qn[c][s], h2 = g2s[c][s][j];
        const float g2 = h2 > 0.0f ? dq * w3 : 0.0f;
        dw3 = fmaf(dq, h2, dw3), db3 += dq, db2 += g2;
        g2s[c][s][j] = g2;
    }
    // column j of W2 (still in LDS), and the first layer's gradient accumulator in place of its weight
    float w2c[H];
#pragma unroll
    for (int i = 0; i < H; ++i) w2c[i] = w2s[1][c][i][j];
    float dw1[DR];
#pragma unroll
    for (int d = 0; d < DR; ++d) dw1[d] = 0.0f;
    for (int d = DR; d < D; ++d) w1s[c][j][d] = 0.0f;
    __syncthreads();
    float dw2[H];
#pragma unroll
    for (int i = 0; i < H; ++i) dw2[i] = 0.0f;
    float db1 = 0.0f;
#pragma unroll 2
    for (int s = s0; s < s1; ++s) {
        const float g2 = g2s[c][s][j];
        float g1a = 0.0f, g1b = 0.0f;
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            const float4 h = *(const float4 *)&h1s[c][s][i], g = *(const float4 *)&g2s[c][s][i];
            dw2[i] = fmaf(g2, h.x, dw2[i]), dw2[i + 1] = fmaf(g2, h.y, dw2[i + 1]), dw2[i + 2] = fmaf(g2, h.z, dw2[i + 2]), dw2[i + 3] = fmaf(g2, h.w, dw2[i + 3]);
            g1a = fmaf(w2c[i], g.x, g1a), g1b = fmaf(w2c[i + 1], g.y, g1b), g1a = fmaf(w2c[i + 2], g.z, g1a), g1b = fmaf(w2c[i + 3], g.w, g1b);
        }
        const float g1 = h1s[c][s][j] > 0.0f ? g1a + g1b : 0.0f;
        db1 += g1;
#pragma unroll
        for (int d = 0; d < DR; d += 4) {
            if (d >= D) break;      // (uniform)
            const float4 v = *(const float4 *)&xo[s][d];
            dw1[d] = fmaf(g1, v.x, dw1[d]), dw1[d + 1] = fmaf(g1, v.y, dw1[d + 1]), dw1[d + 2] = fmaf(g1, v.z, dw1[d + 2]), dw1[d + 3] = fmaf(g1, v.w, dw1[d + 3]);
        }
        for (int d = DR; d < D; ++d) w1s[c][j][d] = fmaf(g1, xo[s][d], w1s[c][j][d]);
    }
    // ---- the workgroup's partial: a net's parameters in order, EXCEPT that the 64 x 64 block is stored transposed (lane j writes
    // d W2[j][i] to [i][j]: coalesced; critic_td_reduce_kernel puts it back)
    if (HV == 2) {      // the second half's sums through LDS (regions no pass reads any more), added by the first
        __syncthreads();
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < H; ++i) w2s[0][c][i][j] = dw2[i];
#pragma unroll
            for (int d = 0; d < DR; ++d) h1s[c][d][j] = dw1[d];
            ps[c][0][j] = db1, ps[c][1][j] = db2, ps[c][2][j] = dw3, ps[c][3][j] = db3;
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < H; ++i) dw2[i] += w2s[0][c][i][j];
#pragma unroll
            for (int d = 0; d < DR; ++d) dw1[d] += h1s[c][d][j];
            db1 += ps[c][0][j], db2 += ps[c][1][j], dw3 += ps[c][2][j], db3 += ps[c][3][j];
        }
    }
    float *out = a.partial + (long)blockIdx.x * (n_params + 2);
    if (live && half == 0) {
        float *o = out + (long)c * npn;
#pragma unroll
        for (int d = 0; d < DR; ++d)
            if (d < D) o[j * D + d] = dw1[d];
        for (int d = DR; d < D; ++d) o[j * D + d] = w1s[c][j][d];
        o += H * D;
        o[j] = db1;
        o += H;
#pragma unroll
        for (int i = 0; i < H; ++i) o[i * H + j] = dw2[i];
        o += H * H;
        o[j] = db2;
        o += H;
        o[j] = dw3;
        o += H;
        if (j == 0) o[0] = db3;
    }
    // loss partial (both critics), ok count
    if (j == 0) lossw[w] = live ? loss : 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[n_params] = HV == 2 ? (lossw[0] + lossw[2]) + (lossw[1] + lossw[3]) : lossw[0] + lossw[1];
        float n = 0.0f;
        for (int s = 0; s < S; ++s) n += okf[s];
        out[n_params + 1] = n;
    }
}

// grad[t] = out_scale / max(1, n_ok) * sum_blocks partial[block][t]  (fp64), loss = sum / max(1, n_ok).  A workgroup takes 64 entries; its
// four wavefronts take a quarter of the blocks each (16 loads in flight per lane) and are added in a fixed order.
__global__ void __launch_bounds__(256) critic_td_reduce_kernel(const float *partial, int n_blocks, int n_params, int D, double out_scale, double *grad, float *loss_out) {
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
    __syncthreads();
    const int p = threadIdx.x & 63, sl = threadIdx.x >> 6, t = blockIdx.x * 64 + p;
    const int per = (n_blocks + 3) / 4, lo = sl * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    double acc = 0.0;
    if (t <= n_params) {
        int b = lo;
        for (; b + 16 <= hi; b += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = partial[(long)(b + k) * stride + t];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += (double)v[k];
        }
        for (; b < hi; ++b) acc += (double)partial[(long)b * stride + t];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && t <= n_params) {
        acc = ((red[p] + red[64 + p]) + red[128 + p]) + red[192 + p];
        if (t < n_params) {
            // where the entry lives in torch's order: the 64 x 64 block of a net comes in transposed
            const int npn = critic_params_per_net(D), net = t / npn, l = t - net * npn, w2 = CRITIC_H * D + CRITIC_H;
            int dst = t;
            if (l >= w2 && l < w2 + CRITIC_H * CRITIC_H) dst = net * npn + w2 + ((l - w2) & 63) * CRITIC_H + ((l - w2) >> 6);
            grad[dst] = acc * out_scale / cnt;
        } else if (loss_out)
            *loss_out = (float)(acc / cnt);
    }
}

// dQ_1/da at (s_b, a_b) for the deterministic policy gradient (what autograd of ContinuousCritic.q1_forward, rlmpc/td3/policies.py:68-76,
// gives): forward of the first critic, then back through the two ReLU layers to the action inputs.  Rows with ok[b] = 0 get 0 (their
// inputs are read as 0, like the torch expression that selects them out).  One workgroup of one wavefront per 4 rows.
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
    constexpr int H = CRITIC_H, S = 4;      // four rows per single-wavefront workgroup: 1024 workgroups at batch 4096
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
    float w2r[H], w2c[H];
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
#pragma unroll 1
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
#pragma unroll 1
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
    // dq/da_u = sum_i W1[i][nx + u] g1_i: lane = (row s, sixteenth k)
    const int s = j >> 4, k = j & 15, b = b0 + s;
    for (int u = 0; u < a.nu; ++u) {
        float v = 0.0f;
        for (int i = 0; i < 4; ++i) v = fmaf(W1[(k * 4 + i) * D + a.nx + u], g1s[s][k * 4 + i], v);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (k == 0 && b < a.B) a.dq_da[(long)b * a.nu + u] = okf[s] != 0.0f ? v : 0.0f;
    }
}

}  // namespace mpcrl
