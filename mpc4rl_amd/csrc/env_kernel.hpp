// env_kernel.hpp — K4 of SURVEY.md §2: the batched cartpole swing-up environment as one launch per call, one lane per environment.
// step: the explicit Euler step of rlmpc/gym/continuous_cartpole/environment.py:105-134 (force = force_mag * a, the "as written"
// accelerations, tau = 0.02), reward x^2 + theta^2 of the NEW state (:193-194), the terminal box test (:136-146) and the step-count
// truncation of gymnasium's TimeLimit.  reset: theta ~ (0.9 + 0.2 u) pi, rest 0 (:178-180) for the masked environments, u drawn by
// the caller (the generator stays the host framework's).  All arithmetic in fp64 like the reference's numpy state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcrl {

struct CartpoleEnvPar {
    double gravity, masscart, masspole, length, force_mag, tau, x_threshold, theta_threshold;
    long max_episode_steps;
};

// OBS: the type observations are stored in — double, or float as the reference's gymnasium environments return them
// (np.array(self.state, dtype=np.float32), environment.py:166,186): the state itself stays fp64 like the reference's numpy state
template <class OBS>
__device__ __forceinline__ void store_obs4(OBS *obs, int i, double a, double b, double c, double d) {
    if constexpr (sizeof(OBS) == 8) {
        reinterpret_cast<double2 *>(obs)[2 * i] = make_double2(a, b);
        reinterpret_cast<double2 *>(obs)[2 * i + 1] = make_double2(c, d);
    } else
        reinterpret_cast<float4 *>(obs)[i] = make_float4((float)a, (float)b, (float)c, (float)d);
}

// one environment's step (shared by env_cartpole_step_kernel and the fused roll-out kernel of td3_kernel.hpp: the same expressions, the same bits)
struct CartpoleStepOut {
    double nx, nxd, nth, nthd, reward;
    bool terminated;
};
__device__ __forceinline__ CartpoleStepOut cartpole_env_step(const CartpoleEnvPar &p, double x, double xd, double th, double thd, double action) {
    const double total_mass = p.masspole + p.masscart, pml = p.masspole * p.length;
    const double force = action * p.force_mag;
    const double c = cos(th), s = sin(th);
    const double temp = (force + pml * (thd * thd) * s) / total_mass;
    const double thacc = (p.gravity * s - c * temp) / (p.length * (4.0 / 3.0 - p.masspole * (c * c) / total_mass));
    const double xacc = temp - pml * thacc * c / total_mass;
    CartpoleStepOut o;
    o.nx = x + p.tau * xd, o.nxd = xd + p.tau * xacc, o.nth = th + p.tau * thd, o.nthd = thd + p.tau * thacc;
    o.reward = o.nx * o.nx + o.nth * o.nth;
    o.terminated = fabs(o.nx) < p.x_threshold && fabs(o.nxd) < 0.1 && fabs(o.nth) < p.theta_threshold && fabs(o.nthd) < 0.1;
    return o;
}

template <class OBS>
__global__ void __launch_bounds__(256) env_cartpole_step_kernel(const CartpoleEnvPar p, int B, double *state, int64_t *steps, const double *action,
                                                                OBS *obs, double *reward, uint8_t *terminated, uint8_t *truncated) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const double2 s01 = reinterpret_cast<const double2 *>(state)[2 * i], s23 = reinterpret_cast<const double2 *>(state)[2 * i + 1];
    const CartpoleStepOut o = cartpole_env_step(p, s01.x, s01.y, s23.x, s23.y, action[i]);
    reinterpret_cast<double2 *>(state)[2 * i] = make_double2(o.nx, o.nxd);
    reinterpret_cast<double2 *>(state)[2 * i + 1] = make_double2(o.nth, o.nthd);
    if (obs) store_obs4(obs, i, o.nx, o.nxd, o.nth, o.nthd);
    const int64_t n = steps[i] + 1;
    steps[i] = n;
    reward[i] = o.reward;
    terminated[i] = o.terminated;
    truncated[i] = n >= p.max_episode_steps;
}

template <class OBS>
__global__ void __launch_bounds__(256) env_cartpole_reset_kernel(int B, double *state, int64_t *steps, const uint8_t *mask, const double *u01, OBS *obs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    if (!mask || mask[i]) {
        reinterpret_cast<double2 *>(state)[2 * i] = make_double2(0.0, 0.0);
        reinterpret_cast<double2 *>(state)[2 * i + 1] = make_double2((0.9 + 0.2 * u01[i]) * 3.141592653589793, 0.0);
        steps[i] = 0;
    }
    if (obs) {
        const double2 a = reinterpret_cast<const double2 *>(state)[2 * i], b = reinterpret_cast<const double2 *>(state)[2 * i + 1];
        store_obs4(obs, i, a.x, a.y, b.x, b.y);
    }
}

// Linear-system environment (rlmpc/gym/linear_system/environment.py:28-58): s+ = A s + B a + [lb + (ub - lb) u01, 0], cost of the NEW
// state = 1/2 s's + 1/2 a'a + 100 per violated side of the observation box.  par: A (row-major 2x2), B (2), lb_noise, ub_noise,
// low (2), high (2).
struct LinearEnvPar {
    double A[4], B[2], lb_noise, ub_noise, low[2], high[2];
};

template <class OBS>
__global__ void __launch_bounds__(256) env_linear_step_kernel(const LinearEnvPar p, int B, double *state, const double *action, const double *u01,
                                                              OBS *obs, double *cost) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const double2 s = reinterpret_cast<const double2 *>(state)[i];
    const double a = action[i];
    const double n0 = p.lb_noise + (p.ub_noise - p.lb_noise) * u01[i];
    const double s0 = (s.x * p.A[0] + s.y * p.A[1]) + a * p.B[0] + n0;
    const double s1 = (s.x * p.A[2] + s.y * p.A[3]) + a * p.B[1];
    reinterpret_cast<double2 *>(state)[i] = make_double2(s0, s1);
    if (obs) {
        if constexpr (sizeof(OBS) == 8)
            reinterpret_cast<double2 *>(obs)[i] = make_double2(s0, s1);
        else
            reinterpret_cast<float2 *>(obs)[i] = make_float2((float)s0, (float)s1);
    }
    const double lower = (p.low[0] - s0 > 0.0 || p.low[1] - s1 > 0.0) ? 1e2 : 0.0;
    const double upper = (s0 - p.high[0] > 0.0 || s1 - p.high[1] > 0.0) ? 1e2 : 0.0;
    cost[i] = 0.5 * (s0 * s0 + s1 * s1) + 0.5 * (a * a) + lower + upper;
}

// The actor's output stage for a batch (round 6): what rlmpc/td3/policies.py:186-213 + MPC.scale_action (rlmpc/mpc/common/mpc.py:290-301) + TD3's
// exploration / target-policy noise do per observation, one lane per instance:
//     ok_i = status_i accepted (0, or also 2) and u0_i finite;   a_i = ok_i ? 2 (u0_i - lo) / (hi - lo) - 1 : 0   (scale = 0: a_i = u0_i)
//     a_i  = clip(a_i + clip(sigma * noise_i, -noise_clip, noise_clip), -1, 1)      (noise = NULL: a_i as it is, unclipped)
// The arithmetic keeps the order and types of the torch expressions it replaces (the scaling in fp64, then float; the noise product, its
// clip, the sum and the final clip in float), so that a loop switched to it reproduces its numbers.
__device__ __forceinline__ float policy_action_one(double u, bool good, double lo, double hi, int scale, const float *noise, float sigma, float noise_clip) {
    float a = !good ? 0.0f : (float)(scale ? 2.0 * ((u - lo) / (hi - lo)) - 1.0 : u);
    if (noise) {
#pragma clang fp contract(off)      // the product and the sum stay two roundings, as in the torch expressions, wherever this is inlined
        float n = sigma * *noise;
        if (noise_clip > 0.0f) n = fminf(fmaxf(n, -noise_clip), noise_clip);
        a = fminf(fmaxf(a + n, -1.0f), 1.0f);
    }
    return a;
}

__global__ void __launch_bounds__(256) policy_action_kernel(const double *u0, const int *status, const float *noise, const double *lo, const double *hi,
                                                            int B, int nu, int scale, float sigma, float noise_clip, int accept2, float *action, uint8_t *ok) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int st = status[i];
    bool good = st == 0 || (accept2 && st == 2);
    for (int j = 0; j < nu; ++j) good = good && isfinite(u0[(long)i * nu + j]);
    for (int j = 0; j < nu; ++j)
        action[(long)i * nu + j] = policy_action_one(u0[(long)i * nu + j], good, lo[j], hi[j], scale, noise ? noise + (long)i * nu + j : nullptr, sigma, noise_clip);
    if (ok) ok[i] = good ? 1 : 0;
}

}  // namespace mpcrl
