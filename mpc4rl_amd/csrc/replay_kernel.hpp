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

}  // namespace mpcrl
