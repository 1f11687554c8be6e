// The roll-out side of one TD3 step after the policy's solve, in ONE launch (round 6): what BatchedTD3._collect_step (mpc4rl_amd/td3.py — the
// batched form of stable_baselines3's collect_rollouts as scripts/cartpole_mpc_as_td3_agent_closed_loop.py:40-67 drives it) does with ~36
// framework launches between the solve and the next one — the actor's output stage with exploration noise (policy_action_kernel), the
// environment step (env_cartpole_step_kernel), the replay row, the flag of the stored iterate, the statistics, the reset of the
// environments that ended (env_cartpole_reset_kernel), the observation and the cold mask of the next solve, the replay position.
// One lane per environment; the arithmetic is that of the kernels it stands for (shared device functions: the same bits).  The write
// position lives on the device: every workgroup reads it when it starts, the LAST one to finish (a ticket) advances it and adds the
// workgroups' statistics in block order (no floating-point atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "env_kernel.hpp"

namespace mpcrl {

struct Td3CollectArgs {
    CartpoleEnvPar par;
    int E;
    double *state;            // [E][4] the environments' states
    int64_t *steps;           // [E]
    const double *u0;         // [E] the policy's solve: control
    const int *status;        // [E]
    const float *eps;         // [E] standard-normal draws (exploration)
    const double *u01;        // [E] uniform draws (resets)
    double lo, hi;
    int scale;
    float sigma;
    double *obs;              // [E][4] in: the observation the solve was at; out: the next one (after resets)
    int32_t *ended;           // [E] out: 1 = the episode ended (the next solve starts that instance cold)
    float *table;             // [cap][E][11]: obs | next obs | action | reward | done
    int cap;
    double reward_scale;
    int64_t *pos;             // [1] the slot to write; advanced by the last workgroup
    uint8_t *iter_ok;         // [cap][E] or nullptr
    int64_t *iter_rows;       // [E] or nullptr: out: the rows of this step's iterates in the caller's tables (pos * E + env)
    double *stats;            // [3]: += sum of rewards, converged solves, episodes ended
    double *partial;          // [n_blocks][3]
    unsigned int *ticket;     // [1], zero before the first launch
};

__global__ void __launch_bounds__(256) td3_cartpole_collect_kernel(const Td3CollectArgs a) {
    __shared__ double red[3][256];
    __shared__ bool last;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t pos = a.pos[0], npos = (pos + 1) % a.cap;
    double s_rew = 0.0, s_conv = 0.0, s_done = 0.0;
    if (i < a.E) {
        const int st = a.status[i];
        const double u = a.u0[i];
        const bool good = (st == 0 || st == 2) && isfinite(u);
        const float act = policy_action_one(u, good, a.lo, a.hi, a.scale, a.eps + i, a.sigma, 0.0f);
        const double2 s01 = reinterpret_cast<const double2 *>(a.state)[2 * i], s23 = reinterpret_cast<const double2 *>(a.state)[2 * i + 1];
        const CartpoleStepOut o = cartpole_env_step(a.par, s01.x, s01.y, s23.x, s23.y, (double)act);
        const int64_t n = a.steps[i] + 1;
        const bool trunc = n >= a.par.max_episode_steps, done = o.terminated || trunc;
        // the replay row
        float *row = a.table + ((long)pos * a.E + i) * 11;
        const double2 o01 = reinterpret_cast<const double2 *>(a.obs)[2 * i], o23 = reinterpret_cast<const double2 *>(a.obs)[2 * i + 1];
        row[0] = (float)o01.x, row[1] = (float)o01.y, row[2] = (float)o23.x, row[3] = (float)o23.y;
        row[4] = (float)o.nx, row[5] = (float)o.nxd, row[6] = (float)o.nth, row[7] = (float)o.nthd;
        row[8] = act, row[9] = (float)(a.reward_scale * o.reward), row[10] = o.terminated ? 1.0f : 0.0f;
        if (a.iter_ok) a.iter_ok[(long)pos * a.E + i] = (good && st == 0) ? 1 : 0;
        if (a.iter_rows) a.iter_rows[i] = pos * a.E + i;
        s_rew = o.reward, s_conv = st == 0 ? 1.0 : 0.0, s_done = done ? 1.0 : 0.0;
        // the environment goes on, or starts again
        double x = o.nx, xd = o.nxd, th = o.nth, thd = o.nthd;
        int64_t cnt = n;
        if (done) x = 0.0, xd = 0.0, th = (0.9 + 0.2 * a.u01[i]) * 3.141592653589793, thd = 0.0, cnt = 0;
        reinterpret_cast<double2 *>(a.state)[2 * i] = make_double2(x, xd);
        reinterpret_cast<double2 *>(a.state)[2 * i + 1] = make_double2(th, thd);
        a.steps[i] = cnt;
        reinterpret_cast<double2 *>(a.obs)[2 * i] = make_double2(x, xd);
        reinterpret_cast<double2 *>(a.obs)[2 * i + 1] = make_double2(th, thd);
        a.ended[i] = done ? 1 : 0;
    }
    red[0][threadIdx.x] = s_rew, red[1][threadIdx.x] = s_conv, red[2][threadIdx.x] = s_done;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x < 3) a.partial[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (int k = 0; k < (int)gridDim.x; ++k) acc += __builtin_nontemporal_load(&a.partial[k * 3 + threadIdx.x]);
        a.stats[threadIdx.x] += acc;
    }
    if (threadIdx.x == 0) a.pos[0] = npos, *a.ticket = 0u;
}

// After the collective, every policy_delay-th update (BatchedTD3._update_post): the policy step from the all-reduced message
//     step = lr * mask * g / max(1, count);  theta += step;  theta' = (1 - tau) theta' + tau theta
// and the Polyak update of the target critics, critic' = (1 - tau) critic' + tau critic — nine framework launches — in one.
__global__ void __launch_bounds__(256) td3_policy_post_kernel(const double *msg, int n_theta, double lr, const double *mask, double tau, double *theta,
                                                              double *theta_target, double *step_out, const float *crit, float *crit_target, int n_crit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_theta) {
        const double cnt = msg[n_theta] > 1.0 ? msg[n_theta] : 1.0;
        const double st = lr * mask[i] * msg[i] / cnt;
        const double th = theta[i] + st;
        theta[i] = th, step_out[i] = st;
        theta_target[i] = theta_target[i] * (1.0 - tau) + tau * th;
    }
    if (i < n_crit) crit_target[i] = crit_target[i] * (float)(1.0 - tau) + (float)tau * crit[i];
}

}  // namespace mpcrl
