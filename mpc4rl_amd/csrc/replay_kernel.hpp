// Replay-side glue of the TD3 loop (round 6): what DeviceReplayBuffer.sample + iterates_of_last_sample (mpc4rl_amd/td3.py — the batched
// form of stable_baselines3's ReplayBuffer.sample as scripts/cartpole_mpc_as_td3_agent_closed_loop.py:47-58 uses it) do with ~22 framework
// launches (gather, integer index arithmetic, masks, float64 copies of the states for the two replay solves), in one launch, one lane per
// sampled transition.  Nothing is computed in floating point: every output is a copy, a widening or an integer expression.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcrl {

struct ReplaySampleArgs {
    const float *table;       // [cap * E][row_len]: obs (nx) | next obs (nx) | action (nu) | reward | done
    int row_len, nx, B, E, cap, steps;
    const int64_t *idx;       // [B] sampled rows (step * E + env)
    const int64_t *pos_t;     // [1] the slot the roll-out writes next
    int exclude;              // != 0: that slot is being written WHILE this batch is drawn (pipelined loop): idx counts the rows of the other cap - 1 slots
    const uint8_t *iter_ok;   // [cap * E] or nullptr: the roll-out solve stored with the transition converged
    float *rows;              // [B][row_len]
    double *obs64, *nxt64;    // [B][nx]
    int64_t *row_s, *row_n;   // [B] rows of the iterate tables to start the solves at obs / next obs from
    int32_t *cold_s, *cold_n; // [B] 1 = that stored iterate is not a starting point
};

__global__ void __launch_bounds__(256) replay_sample_kernel(const ReplaySampleArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.B) return;
    int64_t i = a.idx[b];
    if (a.exclude) {      // row of the k-th slot after the excluded one
        const int64_t st = i / a.E;
        i = ((a.pos_t[0] + 1 + st) % a.cap) * a.E + (i - st * a.E);
    }
    const float *r = a.table + i * a.row_len;
    float *o = a.rows + (long)b * a.row_len;
    for (int k = 0; k < a.row_len; ++k) o[k] = r[k];
    for (int k = 0; k < a.nx; ++k) a.obs64[(long)b * a.nx + k] = (double)r[k], a.nxt64[(long)b * a.nx + k] = (double)r[a.nx + k];
    if (a.iter_ok) {
        // the iterate for next obs: the roll-out solution of the NEXT step of the same environment where that exists and is that state
        // (the episode went on, the slot is neither the one about to be overwritten nor one not yet written) — else the one at obs
        const int64_t step = i / a.E, env = i - step * a.E, nstep = (step + 1) % a.cap;
        const bool cont = r[a.row_len - 1] == 0.0f && nstep != a.pos_t[0] && nstep < a.steps;
        const int64_t n = cont ? nstep * a.E + env : i;
        a.row_s[b] = i, a.row_n[b] = n;
        a.cold_s[b] = a.iter_ok[i] ? 0 : 1, a.cold_n[b] = a.iter_ok[n] ? 0 : 1;
    }
}

// The deterministic policy gradient's contraction (round 6): out[p] = sum_b ok_b sum_u dq_da[b][u] chain_u dpi_dp[b][u][p], out[n_p] = sum_b ok_b
// with chain_u = 2 / (hi_u - lo_u) (the derivative of MPC.scale_action, rlmpc/mpc/common/mpc.py:290-301; 1 without scaling) — the step
// the reference leaves as a stub (rlmpc/td3/policies.py:332).  ~14 framework launches, the batch sum of [4096 x 83] doubles alone 48 us,
// in ONE: every workgroup sums DPG_ROWS rows into its partial; the LAST workgroup to finish (a ticket counter) adds the partials in
// block order — no floating-point atomics, the same bits whatever the scheduling.  Non-finite sensitivities are read as nan_to_num does.
constexpr int DPG_ROWS = 32, DPG_PMAX = 256;

struct DpgArgs {
    const float *dq_da;      // [B][nu]
    const uint8_t *ok;       // [B] or nullptr
    const double *dpi_dp;    // [B][nu][n_p]
    int B, nu, n_p, scale;
    const double *lo, *hi;   // [nu]
    double *partial;         // [n_blocks][n_p + 1]
    unsigned int *ticket;    // [1], zero before the first launch (the kernel leaves it zero)
    double *out;             // [n_p + 1]
};

__device__ inline double nan_to_num_d(double v) {
    return v != v ? 0.0 : (v > 1.7976931348623157e308 ? 1.7976931348623157e308 : (v < -1.7976931348623157e308 ? -1.7976931348623157e308 : v));
}

__global__ void __launch_bounds__(128) dpg_grad_kernel(const DpgArgs a) {
    __shared__ double w[DPG_ROWS][8];      // the rows' weights (nu <= 8), 0 for a row that is left out
    __shared__ double cnt;
    __shared__ bool last;
    __shared__ double fin[4][DPG_PMAX];
    const int b0 = blockIdx.x * DPG_ROWS, P1 = a.n_p + 1;
    for (int e = threadIdx.x; e < DPG_ROWS * a.nu; e += 128) {
        const int r = e / a.nu, u = e - r * a.nu, b = b0 + r;
        const bool ok = b < a.B && (!a.ok || a.ok[b]);
        const double chain = a.scale ? 2.0 / (a.hi[u] - a.lo[u]) : 1.0;
        w[r][u] = ok ? (double)a.dq_da[(long)b * a.nu + u] * chain : 0.0;
    }
    if (threadIdx.x == 0) {
        double n = 0.0;
        for (int r = 0; r < DPG_ROWS; ++r) n += (b0 + r < a.B && (!a.ok || a.ok[b0 + r])) ? 1.0 : 0.0;
        cnt = n;
    }
    __syncthreads();
    // (a row that is left out has weight 0 and is read all the same — nan_to_num makes every entry finite, 0 x finite = 0: no branch,
    // eight loads in flight per lane)
    const int nr = (a.B - b0 < DPG_ROWS ? a.B - b0 : DPG_ROWS) * a.nu;      // (row, control) pairs of this workgroup
    const double *base = a.dpi_dp + (long)b0 * a.nu * a.n_p;
    for (int p = threadIdx.x; p < a.n_p; p += 128) {
        double acc = 0.0;
        int k = 0;
        for (; k + 8 <= nr; k += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = base[(long)(k + q) * a.n_p + p];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fma((&w[0][0])[((k + q) / a.nu) * 8 + (k + q) % a.nu], nan_to_num_d(v[q]), acc);
        }
        for (; k < nr; ++k) acc = fma((&w[0][0])[(k / a.nu) * 8 + k % a.nu], nan_to_num_d(base[(long)k * a.n_p + p]), acc);
        a.partial[(long)blockIdx.x * P1 + p] = acc;
    }
    if (threadIdx.x == 0) a.partial[(long)blockIdx.x * P1 + a.n_p] = cnt;
    // the last workgroup to get here adds the partials up, in block order
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // (a dependent chain of loads over the blocks would cost their latency each: four slices of the blocks per entry, eight loads in flight,
    // the slices added in order)
    const int nb = gridDim.x, per = (nb + 3) / 4;
    for (int p0 = 0; p0 < P1; p0 += DPG_PMAX) {
        const int np = P1 - p0 < DPG_PMAX ? P1 - p0 : DPG_PMAX;
        for (int e = threadIdx.x; e < 4 * np; e += 128) {
            const int sl = e / np, p = p0 + e - sl * np;
            const int lo = sl * per, hi = lo + per < nb ? lo + per : nb;
            double acc = 0.0;
            int k = lo;
            for (; k + 8 <= hi; k += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = a.partial[(long)(k + q) * P1 + p];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q];
            }
            for (; k < hi; ++k) acc += a.partial[(long)k * P1 + p];
            fin[sl][e - sl * np] = acc;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < np; e += 128) a.out[p0 + e] = ((fin[0][e] + fin[1][e]) + fin[2][e]) + fin[3][e];
        __syncthreads();
    }
    if (threadIdx.x == 0) *a.ticket = 0u;
}

}  // namespace mpcrl
