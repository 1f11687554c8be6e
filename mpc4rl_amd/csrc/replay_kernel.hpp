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
    const uint8_t *iter_ok;   // [cap * E] or nullptr: the roll-out solve stored with the transition converged
    float *rows;              // [B][row_len]
    double *obs64, *nxt64;    // [B][nx]
    int64_t *row_s, *row_n;   // [B] rows of the iterate tables to start the solves at obs / next obs from
    int32_t *cold_s, *cold_n; // [B] 1 = that stored iterate is not a starting point
};

__global__ void __launch_bounds__(256) replay_sample_kernel(const ReplaySampleArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.B) return;
    const int64_t i = a.idx[b];
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
constexpr int DPG_ROWS = 32;

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
    for (int p = threadIdx.x; p < a.n_p; p += 128) {
        double acc = 0.0;
        for (int r = 0; r < DPG_ROWS && b0 + r < a.B; ++r)
            for (int u = 0; u < a.nu; ++u) {
                const double wt = w[r][u];
                if (wt != 0.0) acc = fma(wt, nan_to_num_d(a.dpi_dp[((long)(b0 + r) * a.nu + u) * a.n_p + p]), acc);
            }
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
    for (int p = threadIdx.x; p < P1; p += 128) {
        double acc = 0.0;
        for (int k = 0; k < (int)gridDim.x; ++k) acc += __builtin_nontemporal_load(&a.partial[(long)k * P1 + p]);
        a.out[p] = acc;
    }
    if (threadIdx.x == 0) *a.ticket = 0u;
}

}  // namespace mpcrl
