// mpcrl_api.hip — C ABI of libmpcrl_hip.so (include/mpcrl.h): handle management, workspace, kernel launches.
// Everything here is host glue; the arithmetic lives in small_kernel.hpp (cartpole, linear system) and
// chain_kernel.hpp (chain of masses).
#include "mpcrl_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "small_kernel.hpp"
#include "linear_kernel.hpp"
#include "reduce_kernel.hpp"
#include "order_kernel.hpp"
#include "iterate_kernel.hpp"
#include "env_kernel.hpp"
#include "critic_kernel.hpp"
#include "replay_kernel.hpp"
#include "td3_kernel.hpp"

using namespace mpcrl;

namespace {

// switches to the handle's device for the duration of a call and restores the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(dev)            \
    DeviceGuard guard_((dev));    \
    if (!guard_.ok) return MPCRL_E_HIP

// the handle-less library kernels (reduction, environments) launch on the device that OWNS the memory they are given, whatever device
// is current in the calling thread; -1 if the pointer is not device memory
int device_of(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    // device memory, or managed memory (its home device); host-registered / unregistered pointers are refused
    return (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) ? at.device : -1;
}
#define ON_DEVICE_OF(ptr)                       \
    const int dev_of_ = device_of((ptr));       \
    if (dev_of_ < 0) return MPCRL_E_ARG;        \
    ON_DEVICE(dev_of_)

template <class T>
int dev_alloc(T **p, size_t n, int64_t &bytes) {
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) return MPCRL_E_NOMEM;
    bytes += (int64_t)(n * sizeof(T));
    return 0;
}

void copy_or_fill(double *dst, const double *src, int n, int cap, double fill) {
    for (int i = 0; i < cap; ++i) dst[i] = (src && i < n) ? src[i] : fill;
}

int fill_small_spec(const MpcrlProblemSpec &s, SmallSpec &d) {
    const int nw = s.nx + s.nu;
    if (nw > SMALL_MAXNW || s.n_consts > SMALL_MAXC || s.N + 1 > 64 || s.N < 2) return MPCRL_E_ARG;
    std::memset(&d, 0, sizeof(d));
    d.N = s.N, d.np = s.np, d.cost_kind = s.cost_kind, d.rk_steps = s.rk_steps, d.max_iter = s.max_iter;
    d.dT = s.dT, d.gamma = s.gamma, d.h = s.h, d.tol = s.tol;
    copy_or_fill(d.lb0, s.lb0, s.nu, SMALL_MAXNW, -MPCRL_NO_BOUND);
    copy_or_fill(d.ub0, s.ub0, s.nu, SMALL_MAXNW, MPCRL_NO_BOUND);
    copy_or_fill(d.lb, s.lb, nw, SMALL_MAXNW, -MPCRL_NO_BOUND);
    copy_or_fill(d.ub, s.ub, nw, SMALL_MAXNW, MPCRL_NO_BOUND);
    copy_or_fill(d.lbe, s.lbe, s.nx, SMALL_MAXNW, -MPCRL_NO_BOUND);
    copy_or_fill(d.ube, s.ube, s.nx, SMALL_MAXNW, MPCRL_NO_BOUND);
    copy_or_fill(d.zl, s.zl, nw, SMALL_MAXNW, 0.0);
    copy_or_fill(d.zu, s.zu, nw, SMALL_MAXNW, 0.0);
    for (int i = 0; i < SMALL_MAXNW; ++i) d.soft[i] = (s.soft && i < nw) ? s.soft[i] : 0;
    copy_or_fill(d.consts, s.consts, s.n_consts, SMALL_MAXC, 0.0);
    return 0;
}

int fill_large_spec(const MpcrlProblemSpec &s, LargeSpec &d) {
    const int nw = s.nx + s.nu;
    if (nw > LARGE_MAXNW || s.nu > 4 || s.N + 1 > 64 || s.N < 1 || s.rk_steps < 1 || s.rk_steps > 2) return MPCRL_E_ARG;
    if (s.soft)
        for (int i = 0; i < nw; ++i)
            if (s.soft[i]) return MPCRL_E_ARG;   // hard bounds only in the large kernel
    std::memset(&d, 0, sizeof(d));
    d.N = s.N, d.np = s.np, d.cost_kind = s.cost_kind, d.rk_steps = s.rk_steps, d.max_iter = s.max_iter;
    d.dT = s.dT, d.gamma = s.gamma, d.h = s.h, d.tol = s.tol;
    copy_or_fill(d.lb0, s.lb0, s.nu, 4, -MPCRL_NO_BOUND);
    copy_or_fill(d.ub0, s.ub0, s.nu, 4, MPCRL_NO_BOUND);
    copy_or_fill(d.lb, s.lb, nw, LARGE_MAXNW, -MPCRL_NO_BOUND);
    copy_or_fill(d.ub, s.ub, nw, LARGE_MAXNW, MPCRL_NO_BOUND);
    copy_or_fill(d.lbe, s.lbe, s.nx, LARGE_MAXNW, -MPCRL_NO_BOUND);
    copy_or_fill(d.ube, s.ube, s.nx, LARGE_MAXNW, MPCRL_NO_BOUND);
    return 0;
}

// n_mass (3 .. 7, a free integer in the reference: rlmpc/mpc/chain_mass/ocp_utils.py:344-350) -> the translation unit of that chain size
const MpcrlChainEntry *chain_entry(int n_mass) {
    switch (n_mass) {
        case 3: return mpcrl_chain_entry_3();
        case 4: return mpcrl_chain_entry_4();
        case 5: return mpcrl_chain_entry_5();
        case 6: return mpcrl_chain_entry_6();
        case 7: return mpcrl_chain_entry_7();
        default: return nullptr;
    }
}

// Time-sliced launch (small_solve_sliced_kernel): ipw + 1 instances per wavefront, ipw of them advancing per round.  Such a
// wavefront lives longer than (ipw + 1) / ipw plain lifetimes — measured on cartpole N = 20 at the end of round 2: 1.62 (818 k vs
// 1 328 k cycles: write-outs inside the loop, 3 % for mixing QPs of instances one SQP iteration apart, and more register spill traffic
// than the plain kernel) — so it pays only when it saves enough rounds of wavefronts on the chip's SIMDs: 4096 instances are 2 rounds
// of 3-instance wavefronts or 1 round of 4-instance ones (5.44 -> 6.36 M solves/s); 32768 are 11 vs 8 rounds (plain wins).
//
// That rule counts wavefronts, not work.  A time-sliced wavefront lasts for the SUM of its instances' iterations and, at one round,
// nothing evens the sums out between SIMDs; plain wavefronts last for the MAXIMUM over three neighbours of the packing order and the
// dispatcher hands the second round to whichever SIMD frees up first.  On batches of one difficulty the sliced launch wins (cartpole
// 4096 states of the benchmark box: 0.56 vs 0.60 ms); on replay samples along closed-loop swing-ups (SQP iterations 2..15) the plain
// one does (0.47 vs 0.65 ms, profiles/r03_replay_cold_solve.txt).  Which of the two a caller's batches are cannot be read off the
// flags, so in automatic mode a handle whose batch size passes the rule times the two shapes against each other on its own calls.
enum { TUNE_PERIOD = 64, TUNE_SLICED = 0, TUNE_PLAIN = 1 };

template <class M>
bool sliced_candidate(const MpcrlSolver *h, int flags, long *waves, bool *by_rule) {
    if constexpr (M::HAS_SOFT) {
        return false;
    } else {
        const int lpi = h->N + 1, ipw = std::min(64 / lpi, M::MAX_IPW);
        const int ips = std::min(64 / lpi, M::MAX_IPW - 1), q = ips + 1;
        const bool legal = ips >= 1 && 64 - ips * lpi >= 1 && !(flags & MPCRL_RTI);
        const long waves3 = (h->B + ipw - 1) / ipw, waves4 = (h->B + q - 1) / q;
        const long rounds3 = (waves3 + h->n_simd - 1) / h->n_simd, rounds4 = (waves4 + h->n_simd - 1) / h->n_simd;
        if (waves) *waves = waves4;
        if (by_rule) *by_rule = 162 * rounds4 <= 100 * rounds3;
        return legal;
    }
}

// the shape the tuner currently prefers: plain only once both have been timed and the plain kernel was faster by more than 20 %.
// The margin is what whole steps asked for.  The probes time the solve kernel alone; around it the plain launch carries the
// packing-order kernel (16 us at 4096 instances) and, inside the TD3 loop's graphs, loses more than that: on the benchmark's early
// replay rows the plain kernel probes 13 % faster (0.54 vs 0.63 ms) and the closed-loop step is 2 % SLOWER with it (2.06 vs 2.02 ms);
// on later-training replay rows it probes 27 % faster and the whole solve call is 23 % faster (profiles/r03_replay_cold_solve.txt).
constexpr float TUNE_MARGIN = 0.80f;
inline int tuned_best(const MpcrlSolver::Tuner &t) {
    return (t.ms[0] > 0.f && t.ms[1] > 0.f && t.ms[TUNE_PLAIN] < TUNE_MARGIN * t.ms[TUNE_SLICED]) ? TUNE_PLAIN : TUNE_SLICED;
}

// shape of call number `call` of a tuned handle: calls 1 and 2 of every TUNE_PERIOD are the timed probes (call 0 of a fresh handle
// pays module load and cold caches and is not timed)
inline int tuned_shape(const MpcrlSolver::Tuner &t, bool *timed) {
    const unsigned ph = t.calls % TUNE_PERIOD;
    if (timed) *timed = ph == 1 || ph == 2;
    return ph == 1 ? TUNE_SLICED : (ph == 2 ? TUNE_PLAIN : tuned_best(t));
}
inline void tuner_harvest(MpcrlSolver::Tuner &t) {   // probes that have finished since the last look (never waits)
    for (int m = 0; m < 2; ++m)
        if (t.pending[m] && hipEventQuery(t.ev[m][1]) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.ev[m][0], t.ev[m][1]) == hipSuccess && ms > 0.f) t.ms[m] = ms;
            t.pending[m] = false;
        }
}

template <class M>
bool plan_time_sliced(const MpcrlSolver *h, int flags, long *waves) {
    bool by_rule = false;
    if (!sliced_candidate<M>(h, flags, waves, &by_rule)) return false;
    if (h->slice_mode != 0) return h->slice_mode > 0;
    return by_rule && tuned_shape(h->tune[(flags & MPCRL_COLD) ? 0 : 1], nullptr) == TUNE_SLICED;
}

template <class M>
int launch_small(MpcrlSolver *h, const SmallArgs &a, hipStream_t st) {
    const int lpi = h->N + 1, ipw = std::min(64 / lpi, M::MAX_IPW);
    const int blocks = (h->B + ipw - 1) / ipw;
    if constexpr (std::is_same<M, CartpoleDev>::value) {
        if (a.flags & MPCRL_EXACT_QP) {   // test-only: the exact-QP instantiation of the plain solve kernel, then the shipped sensitivity pass
            hipLaunchKernelGGL(small_solve_kernel<CartpoleDevExact>, dim3(blocks), dim3(64), 0, st, h->small, a);
            if (a.flags & (MPCRL_SENS_V | MPCRL_SENS_PI)) hipLaunchKernelGGL(small_sens_kernel<M>, dim3(blocks), dim3(64), 0, st, h->small, a);
            HIP_OK(hipGetLastError());
            return 0;
        }
    }
    bool sliced = false;
    int timed_shape = -1;
    if constexpr (!M::HAS_SOFT) {
        long waves4 = 0;
        bool by_rule = false;
        if (sliced_candidate<M>(h, a.flags, &waves4, &by_rule)) {
            if (h->slice_mode != 0) {
                sliced = h->slice_mode > 0;
            } else if (by_rule) {
                // No event calls while the stream is being captured into a graph (they would end the capture): the graph gets the
                // shape preferred so far, whatever the query promised.
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                const bool capturing = hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
                MpcrlSolver::Tuner &t = h->tune[(a.flags & MPCRL_COLD) ? 0 : 1];
                if (capturing) {
                    sliced = tuned_best(t) == TUNE_SLICED;
                } else {
                    tuner_harvest(t);
                    bool timed = false;
                    const int probe = tuned_shape(t, &timed);
                    const int shape = h->planned >= 0 ? h->planned : probe;   // a promise made before the harvest above stands
                    t.calls++;
                    sliced = shape == TUNE_SLICED;
                    if (timed && shape == probe && !t.pending[shape]) {
                        if (!t.ev[0][0])
                            for (auto &pair : t.ev)
                                for (auto &e : pair) HIP_OK(hipEventCreate(&e));
                        timed_shape = shape;
                        HIP_OK(hipEventRecord(t.ev[shape][0], st));
                    }
                }
            }
        }
        h->planned = -1;
        if (sliced) {
            const bool lds = sliced_parks_in_lds<M>(h->N), warm = !(a.flags & MPCRL_COLD);
            if (lds && !warm) hipLaunchKernelGGL((small_solve_sliced_kernel<M, true, false>), dim3((unsigned)waves4), dim3(64), 0, st, h->small, a);
            if (lds && warm) hipLaunchKernelGGL((small_solve_sliced_kernel<M, true, true>), dim3((unsigned)waves4), dim3(64), 0, st, h->small, a);
            if (!lds && !warm) hipLaunchKernelGGL((small_solve_sliced_kernel<M, false, false>), dim3((unsigned)waves4), dim3(64), 0, st, h->small, a);
            if (!lds && warm) hipLaunchKernelGGL((small_solve_sliced_kernel<M, false, true>), dim3((unsigned)waves4), dim3(64), 0, st, h->small, a);
        }
    }
    bool lq = false;
    if constexpr (std::is_same<M, LinearDev>::value) {
        // the linear-system model: several stages per lane, four (eight) instances per wavefront each in a DPP row (half row) of its own
        // (linear_kernel.hpp) unless switched off (MPCRL_LINEAR_SPL=1 at mpcrl_create: one stage per lane, small_solve_kernel) or the
        // horizon leaves it no advantage.  Three stages per lane while they fit a row (N <= 47); beyond that FOUR (N <= 63 = the longest
        // horizon mpcrl_create accepts: 16 lanes again).  Round 5 packed those horizons in segments of 17-22 lanes with __shfl
        // (lq_solve_kernel<3, 0>): 1.31-1.54 ms per 4096 solves at N = 48..63 against 0.99-1.20 of the one-stage kernels and 0.69-0.74 of
        // <4, 16> (profiles/r06_lq_long_horizons.txt) — gone.
        const int lpi3 = lq_lanes_per_instance<3>(h->N), rl = lpi3 <= 8 ? 8 : 16, ipw3 = 64 / rl;
        static_assert(LinearDev::HAS_SOFT, "the lq branch relies on `sliced` staying false for this model (no time-sliced launch with soft bounds)");
        if (!sliced && h->linear_spl != 1 && ipw3 > ipw && h->N + 1 <= 64) {
            lq = true;
            const dim3 grid((unsigned)((h->B + ipw3 - 1) / ipw3));
            if (lpi3 > 16)
                hipLaunchKernelGGL((lq_solve_kernel<4, 16>), grid, dim3(64), 0, st, h->small, a);
            else if (rl == 16)
                hipLaunchKernelGGL((lq_solve_kernel<3, 16>), grid, dim3(64), 0, st, h->small, a);
            else
                hipLaunchKernelGGL((lq_solve_kernel<3, 8>), grid, dim3(64), 0, st, h->small, a);
        }
    }
    if (!sliced && !lq) hipLaunchKernelGGL(small_solve_kernel<M>, dim3(blocks), dim3(64), 0, st, h->small, a);
    HIP_OK(hipGetLastError());
    if (timed_shape >= 0) {
        MpcrlSolver::Tuner &t = h->tune[(a.flags & MPCRL_COLD) ? 0 : 1];
        HIP_OK(hipEventRecord(t.ev[timed_shape][1], st));
        t.pending[timed_shape] = true;
    }
#if !MPCRL_FUSE_SENS
    if ((a.flags & (MPCRL_SENS_V | MPCRL_SENS_PI)) && !lq) {   // (lq_solve_kernel runs the pass itself)
        hipLaunchKernelGGL(small_sens_kernel<M>, dim3(blocks), dim3(64), 0, st, h->small, a);
        HIP_OK(hipGetLastError());
    }
#endif
    return 0;
}

// writes the cold iterate (MPC.reset) into the stored-iterate arrays; x0 may be null (x_k = 0)
int fill_cold_iterate(MpcrlSolver *h, const double *x0, bool primal, hipStream_t st) {
    const long n = (long)h->B * 10 * (h->N + 1) * (h->nx + h->nu);
    hipLaunchKernelGGL(cold_iterate_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, st, x0, h->B, h->N, h->nx, h->nu,
                       primal ? h->X : nullptr, primal ? h->U : nullptr, primal ? h->PI : nullptr, h->BND);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

#ifdef MPCRL_PROFILE_PHASES
int mpcrl_debug_phases(unsigned long long *out, int reset) {   // summed over the translation units (each has its own counters)
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {0};
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof(z)));
    }
    for (int n = 3; n <= 7; ++n) {
        unsigned long long t[16];
        if (!chain_entry(n)->debug_phases) continue;
        const int rc = chain_entry(n)->debug_phases(t, reset);
        if (rc) return rc;
        for (int i = 0; i < 16; ++i) out[i] += t[i];
    }
    return 0;
}
#endif

int mpcrl_version(void) { return MPCRL_ABI_VERSION; }

int mpcrl_create(const MpcrlProblemSpec *spec, int batch, int device, mpcrl_handle *out) {
    if (!spec || !out || batch <= 0) return MPCRL_E_ARG;
    ON_DEVICE(device);
    MpcrlSolver *h = new (std::nothrow) MpcrlSolver();
    if (!h) return MPCRL_E_NOMEM;
    h->model = spec->model, h->B = batch, h->device = device, h->nx = spec->nx, h->nu = spec->nu, h->np = spec->np, h->N = spec->N;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_simd = 4 * cus;
        const char *e = std::getenv("MPCRL_TIME_SLICE");   // tests, profiling: same as mpcrl_set_launch_mode
        if (e && *e) h->slice_mode = (*e == '0') ? -1 : 1;
        const char *l = std::getenv("MPCRL_LINEAR_SPL");   // stages per lane of the linear-system solve kernel: 3 (default) or 1
        if (l && *l == '1') h->linear_spl = 1;
    }
    int rc = 0;
    switch (spec->model) {
        case MPCRL_MODEL_CARTPOLE:
            if (spec->nx != CartpoleDev::NX || spec->nu != CartpoleDev::NU || spec->np != CartpoleDev::NP) rc = MPCRL_E_MODEL;
            break;
        case MPCRL_MODEL_LINEAR:
            if (spec->nx != LinearDev::NX || spec->nu != LinearDev::NU || spec->np != LinearDev::NP) rc = MPCRL_E_MODEL;
            break;
        case MPCRL_MODEL_CHAIN:
            h->is_large = true;
            h->n_mass = (spec->nx / 3 - 1) / 2 + 2;
            if (spec->nu != 3 || h->n_mass < 3 || h->n_mass > 7 || spec->nx != (2 * (h->n_mass - 2) + 1) * 3 || spec->n_consts != spec->nx)
                rc = MPCRL_E_MODEL;
            else if (spec->np != chain_entry(h->n_mass)->np)
                rc = MPCRL_E_MODEL;
            else if (spec->N >= 1 && spec->N < 64 && !chain_entry(h->n_mass)->fits(spec->N))
                rc = MPCRL_E_ARG;   // the horizon's trajectories / tables do not fit the LDS of one workgroup at this chain size
            break;
        default: rc = MPCRL_E_MODEL;
    }
    if (!rc) rc = h->is_large ? fill_large_spec(*spec, h->large) : fill_small_spec(*spec, h->small);
    if (!rc && !h->is_large)   // L1-soft bounds only where the model's kernel carries slack state (models_dev.hpp soft_coord)
        for (int i = 0; i < spec->nx + spec->nu; ++i)
            if (h->small.soft[i] && !(spec->model == MPCRL_MODEL_LINEAR ? LinearDev::soft_coord(i) : CartpoleDev::soft_coord(i))) rc = MPCRL_E_MODEL;
    if (rc) {
        delete h;
        return rc;
    }
    const size_t B = batch, N = spec->N, nx = spec->nx, nu = spec->nu, nw = nx + nu;
    rc = dev_alloc(&h->X, B * (N + 1) * nx, h->bytes);
    if (!rc) rc = dev_alloc(&h->U, B * N * nu, h->bytes);
    if (!rc) rc = dev_alloc(&h->PI, B * N * nx, h->bytes);
    if (!rc) rc = dev_alloc(&h->BND, B * 10 * (N + 1) * nw, h->bytes);
    if (!rc) rc = dev_alloc(&h->RES, B * 4, h->bytes);
    if (!rc) rc = dev_alloc(&h->LAG, B, h->bytes);
    if (!rc) rc = dev_alloc(&h->theta, B * (size_t)spec->np, h->bytes);
    if (!rc) rc = dev_alloc(&h->perm, B, h->bytes);
    if (!rc) rc = dev_alloc(&h->cold_mask, B, h->bytes);
    if (!rc) rc = dev_alloc(&h->order_state, 4, h->bytes);
    if (!rc && h->is_large) {
        h->ws_stride = chain_entry(h->n_mass)->ws_doubles(spec->N);
        rc = dev_alloc(&h->ws, B * h->ws_stride, h->bytes);
        if (!rc) rc = dev_alloc(&h->consts_dev, (size_t)spec->n_consts, h->bytes);
        if (!rc) {
            if (hipMemcpy(h->consts_dev, spec->consts, spec->n_consts * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = MPCRL_E_HIP;
            h->large.consts = h->consts_dev;
        }
    }
    // parameters 0, residuals 0, and the stored iterate = the cold iterate with x_k = 0: nothing the handle owns is ever read
    // uninitialised (get_iterate / set_iterate before the first solve)
    if (!rc && hipMemset(h->theta, 0, B * (size_t)spec->np * sizeof(double)) != hipSuccess) rc = MPCRL_E_HIP;
    if (!rc && hipMemset(h->RES, 0, B * 4 * sizeof(double)) != hipSuccess) rc = MPCRL_E_HIP;
    if (!rc && hipMemset(h->LAG, 0, B * sizeof(double)) != hipSuccess) rc = MPCRL_E_HIP;
    if (!rc) {   // "no spread known": a cheap-form order launch that runs before any refresh (a captured first call that was never
                 // replayed) then finds one crowded bucket = the identity order, never garbage
        const double os0[4] = {-1.0, 0.0, 0.0, 0.0};
        if (hipMemcpy(h->order_state, os0, sizeof(os0), hipMemcpyHostToDevice) != hipSuccess) rc = MPCRL_E_HIP;
    }
    if (!rc) rc = fill_cold_iterate(h, nullptr, true, nullptr);
    if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = MPCRL_E_HIP;
    if (rc) {
        mpcrl_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int mpcrl_destroy(mpcrl_handle h) {
    if (!h) return MPCRL_E_ARG;
    DeviceGuard guard_(h->device);
    for (double *p : {h->X, h->U, h->PI, h->BND, h->RES, h->LAG, h->theta, h->ws, h->consts_dev})
        if (p) (void)hipFree(p);
    if (h->perm) (void)hipFree(h->perm);
    if (h->cold_mask) (void)hipFree(h->cold_mask);
    if (h->order_state) (void)hipFree(h->order_state);
    for (auto &t : h->tune)
        for (auto &pair : t.ev)
            for (auto &e : pair)
                if (e) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

int64_t mpcrl_workspace_bytes(mpcrl_handle h) { return h ? h->bytes : 0; }

int mpcrl_set_theta(mpcrl_handle h, const double *theta, int n_theta, int per_instance, void *stream) {
    if (!h || !theta || n_theta != h->np) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    const size_t n = (size_t)h->np * (per_instance ? h->B : 1);
    HIP_OK(hipMemcpyAsync(h->theta, theta, n * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->theta_stride = per_instance ? h->np : 0;
    return 0;
}

int mpcrl_set_gamma(mpcrl_handle h, double gamma) {
    if (!h) return MPCRL_E_ARG;
    h->small.gamma = gamma, h->large.gamma = gamma;
    return 0;
}

int mpcrl_set_options(mpcrl_handle h, double tol, int max_iter) {
    if (!h) return MPCRL_E_ARG;
    if (tol > 0) h->small.tol = tol, h->large.tol = tol;
    if (max_iter >= 0) h->small.max_iter = max_iter, h->large.max_iter = max_iter;
    return 0;
}

int mpcrl_set_exit_rule(mpcrl_handle h, int window, double factor) {
    if (!h || window < 0 || window > 255 || !(factor > 0.0 && factor <= 1.0)) return MPCRL_E_ARG;
    h->small.exit_window = window, h->small.exit_factor = factor;
    h->large.exit_window = window, h->large.exit_factor = factor;
    return 0;
}

int mpcrl_set_order(mpcrl_handle h, const int32_t *perm, void *stream) {
    if (!h) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    if (perm) HIP_OK(hipMemcpyAsync(h->perm, perm, (size_t)h->B * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->have_perm = perm != nullptr;
    return 0;
}

int mpcrl_set_cold_mask(mpcrl_handle h, const int32_t *mask, void *stream) {
    if (!h) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    if (mask) HIP_OK(hipMemcpyAsync(h->cold_mask, mask, (size_t)h->B * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->have_cold_mask = mask != nullptr;
    return 0;
}

int mpcrl_query_time_sliced(mpcrl_handle h, int flags, void *stream) {
    if (!h) return MPCRL_E_ARG;
    if (h->is_large) return 0;
    ON_DEVICE(h->device);
    if (!h->have_iterate) flags |= MPCRL_COLD;
    {   // a solve launched into a stream capture times nothing and takes the shape preferred so far (launch_small): answer with that
        // shape, not with the probe shape the call counter would select, and promise nothing
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool query_failed = hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess;
        if (query_failed) (void)hipGetLastError();   // (not sticky for the caller's next HIP call)
        const bool capturing = query_failed || cap != hipStreamCaptureStatusNone;
        if (capturing && h->slice_mode == 0) {
            bool by_rule = false, cand = false;
            if (h->model == MPCRL_MODEL_CARTPOLE) cand = sliced_candidate<CartpoleDev>(h, flags, nullptr, &by_rule);
            if (h->model == MPCRL_MODEL_LINEAR) cand = sliced_candidate<LinearDev>(h, flags, nullptr, &by_rule);
            h->planned = -1;
            return (cand && by_rule && tuned_best(h->tune[(flags & MPCRL_COLD) ? 0 : 1]) == TUNE_SLICED) ? 1 : 0;
        }
    }
    bool sliced = false, tuned = false, by_rule = false;
    switch (h->model) {
        case MPCRL_MODEL_CARTPOLE:
            sliced = plan_time_sliced<CartpoleDev>(h, flags, nullptr);
            tuned = h->slice_mode == 0 && sliced_candidate<CartpoleDev>(h, flags, nullptr, &by_rule) && by_rule;
            break;
        case MPCRL_MODEL_LINEAR:
            sliced = plan_time_sliced<LinearDev>(h, flags, nullptr);
            tuned = h->slice_mode == 0 && sliced_candidate<LinearDev>(h, flags, nullptr, &by_rule) && by_rule;
            break;
        default: return 0;
    }
    h->planned = tuned ? (sliced ? TUNE_SLICED : TUNE_PLAIN) : -1;   // the next mpcrl_solve keeps this promise
    return sliced ? 1 : 0;
}

int mpcrl_set_launch_mode(mpcrl_handle h, int mode) {
    if (!h || mode < -1 || mode > 1) return MPCRL_E_ARG;
    h->slice_mode = mode, h->planned = -1;
    return 0;
}

int mpcrl_get_launch_times(mpcrl_handle h, int warm, double *sliced_ms, double *plain_ms) {
    if (!h || !sliced_ms || !plain_ms) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    MpcrlSolver::Tuner &t = h->tune[warm ? 1 : 0];
    tuner_harvest(t);
    *sliced_ms = t.ms[TUNE_SLICED], *plain_ms = t.ms[TUNE_PLAIN];
    return tuned_best(t);
}

int mpcrl_auto_order(mpcrl_handle h, const double *x0, void *stream) {
    if (!h || !x0) return MPCRL_E_ARG;
    if (h->B > ORDER_MAX) return MPCRL_E_ARG;   // larger batches: build the permutation outside and pass it to mpcrl_set_order
    if (h->is_large || 64 / (h->N + 1) < 2) {   // one instance per wavefront (chain; N + 1 > 32 stages): nothing shares a lock step
        h->have_perm = false;
        return 0;
    }
    ON_DEVICE(h->device);
    // the coordinate and range the batch is bucketed along are found from scratch on the first call and every 32nd one (never inside a
    // stream capture once they exist: a graph replays the cheap form)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    // (a first call inside a capture refreshes as well — the graph then always does — but does not count: the state is only known to
    // be valid once a refresh launch was issued for real)
    const int refresh = (h->order_calls == 0 || (!capturing && h->order_calls % 32 == 0)) ? 1 : 0;
    if (!(capturing && h->order_calls == 0)) h->order_calls++;
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(ORDER_NT), 0, (hipStream_t)stream, x0, h->B, h->nx, h->perm, h->order_state, refresh);
    HIP_OK(hipGetLastError());
    h->have_perm = true;
    return 0;
}

int mpcrl_set_bounds(mpcrl_handle h, int which, const double *lb, const double *ub) {
    if (!h || !lb || !ub || which < MPCRL_BOUNDS_U0 || which > MPCRL_BOUNDS_TERMINAL) return MPCRL_E_ARG;
    const int nw = h->nx + h->nu;
    const int n = which == MPCRL_BOUNDS_U0 ? h->nu : (which == MPCRL_BOUNDS_STAGE ? nw : h->nx);
    for (int i = 0; i < n; ++i)
        if (!(lb[i] <= ub[i])) return MPCRL_E_ARG;
    double *dl, *du;
    if (h->is_large)
        dl = which == MPCRL_BOUNDS_U0 ? h->large.lb0 : (which == MPCRL_BOUNDS_STAGE ? h->large.lb : h->large.lbe),
        du = which == MPCRL_BOUNDS_U0 ? h->large.ub0 : (which == MPCRL_BOUNDS_STAGE ? h->large.ub : h->large.ube);
    else
        dl = which == MPCRL_BOUNDS_U0 ? h->small.lb0 : (which == MPCRL_BOUNDS_STAGE ? h->small.lb : h->small.lbe),
        du = which == MPCRL_BOUNDS_U0 ? h->small.ub0 : (which == MPCRL_BOUNDS_STAGE ? h->small.ub : h->small.ube);
    if (which == MPCRL_BOUNDS_STAGE && !h->is_large)   // a soft bound must stay a two-sided bound (its slack rows exist for both sides)
        for (int i = 0; i < n; ++i)
            if (h->small.soft[i] && (lb[i] <= -MPCRL_NO_BOUND * 0.1 || ub[i] >= MPCRL_NO_BOUND * 0.1)) return MPCRL_E_ARG;
    for (int i = 0; i < n; ++i) dl[i] = lb[i], du[i] = ub[i];
    return 0;
}

int mpcrl_reset(mpcrl_handle h, const double *x0, void *stream) {
    if (!h) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    // the next solve builds the cold iterate itself from ITS x0 (MPCRL_COLD); the stored arrays are set to the same state so
    // that get_iterate / set_iterate after a reset see the cold iterate and not the previous solution
    int rc = fill_cold_iterate(h, x0, true, (hipStream_t)stream);
    h->have_iterate = false, h->dual_cold = false;
    return rc;
}

int mpcrl_solve(mpcrl_handle h, const double *x0, const double *u0_fixed, int flags, double *u0_out, double *V, double *dV_dp,
                double *dpi_dp, int32_t *status, int32_t *iters, void *stream) {
    if (!h || !x0 || !u0_out || !V || !status) return MPCRL_E_ARG;
    if ((flags & MPCRL_SENS_V) && !dV_dp) return MPCRL_E_ARG;
    if ((flags & MPCRL_SENS_PI) && !dpi_dp) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    // (test-only flag: cartpole through its own kernel instantiation, chain and lq_solve_kernel through a wave-uniform switch; not for RTI calls,
    // not for the one-stage linear kernels)
    if ((flags & MPCRL_EXACT_QP) && ((flags & MPCRL_RTI) || (h->model == MPCRL_MODEL_LINEAR && h->linear_spl == 1))) return MPCRL_E_ARG;
    if (!h->have_iterate) flags |= MPCRL_COLD;
    if (h->dual_cold) flags |= MPCRL_COLD_DUAL;
    SmallArgs a;
    a.B = h->B, a.flags = flags, a.theta_stride = h->theta_stride, a.perm = h->have_perm ? h->perm : nullptr;
    a.cold = (h->have_cold_mask && h->have_iterate) ? h->cold_mask : nullptr;   // one-shot: consumed by this solve
    h->have_cold_mask = false;
    a.x0 = x0, a.u0fix = u0_fixed, a.theta = h->theta;
    a.X = h->X, a.U = h->U, a.PI = h->PI, a.BND = h->BND, a.RES = h->RES, a.LAG = h->LAG;
    a.u0_out = u0_out, a.V = V, a.dV = (flags & MPCRL_SENS_V) ? dV_dp : nullptr, a.dpi = (flags & MPCRL_SENS_PI) ? dpi_dp : nullptr;
    a.status = status, a.iters = iters;
    // chain: sens_out stores only the entries of solved instances that have a gradient; the small models' sensitivity kernel writes its rows in full
    if (h->is_large && a.dV) HIP_OK(hipMemsetAsync(dV_dp, 0, (size_t)h->B * h->np * sizeof(double), st));
    if (h->is_large && a.dpi) HIP_OK(hipMemsetAsync(dpi_dp, 0, (size_t)h->B * h->nu * h->np * sizeof(double), st));
    int rc;
    if (h->is_large) {
        LargeArgs la;
        la.B = a.B, la.flags = a.flags, la.theta_stride = a.theta_stride, la.perm = a.perm, la.cold = a.cold, la.x0 = a.x0, la.u0fix = a.u0fix, la.theta = a.theta;
        la.X = a.X, la.U = a.U, la.PI = a.PI, la.BND = a.BND, la.RES = a.RES, la.LAG = a.LAG, la.ws = nullptr, la.ws_stride = 0;
        la.u0_out = a.u0_out, la.V = a.V, la.dV = a.dV, la.dpi = a.dpi, la.status = a.status, la.iters = a.iters;
        rc = chain_entry(h->n_mass)->launch(h, la, st);
    } else
        switch (h->model) {
            case MPCRL_MODEL_CARTPOLE: rc = launch_small<CartpoleDev>(h, a, st); break;
            case MPCRL_MODEL_LINEAR: rc = launch_small<LinearDev>(h, a, st); break;
            default: rc = MPCRL_E_MODEL;
        }
    // (a solve that left the bound planes alone: the next one must not start its interior point from them)
    if (!rc) h->have_iterate = true, h->dual_cold = (flags & MPCRL_NO_BND_STORE) != 0;
    return rc;
}

int mpcrl_weighted_grad_sum(const double *grad, int64_t ld, const double *weight, int rows, int n, double *out, void *stream) {
    if (!grad || !out || rows < 0 || n < 1 || ld < n) return MPCRL_E_ARG;
    ON_DEVICE_OF(out);
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipMemsetAsync(out, 0, (size_t)(n + 2) * sizeof(double), st));
    if (rows == 0) return 0;
    if (n <= REDUCE_MAXN_ROWPAR) {
        const int blocks = std::min(256, (rows + 255) / 256);
        hipLaunchKernelGGL(grad_reduce_rows_kernel, dim3(blocks), dim3(256), 0, st, grad, (long)ld, weight, rows, n, out);
    } else {
        const int ysplit = std::max(1, std::min(64, rows / 64));
        hipLaunchKernelGGL(grad_reduce_cols_kernel, dim3((n + 255) / 256, ysplit), dim3(256), 0, st, grad, (long)ld, weight, rows, n, out);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_env_cartpole_step(const double *par, int B, double *state, int64_t *steps, const double *action, void *obs, int obs_f32,
                            double *reward, uint8_t *terminated, uint8_t *truncated, void *stream) {
    if (!par || B < 0 || !state || !steps || !action || !reward || !terminated || !truncated) return MPCRL_E_ARG;
    if (B == 0) return 0;
    ON_DEVICE_OF(state);
    CartpoleEnvPar p;
    p.gravity = par[0], p.masscart = par[1], p.masspole = par[2], p.length = par[3], p.force_mag = par[4], p.tau = par[5];
    p.x_threshold = par[6], p.theta_threshold = par[7], p.max_episode_steps = (long)par[8];
    if (obs_f32)
        hipLaunchKernelGGL(env_cartpole_step_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, B, state, steps, action,
                           (float *)obs, reward, terminated, truncated);
    else
        hipLaunchKernelGGL(env_cartpole_step_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, B, state, steps, action,
                           (double *)obs, reward, terminated, truncated);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_policy_action(const double *u0, const int32_t *status, const float *noise, const double *lo, const double *hi, int B, int nu, int scale,
                        double sigma, double noise_clip, int accept_status2, float *action, uint8_t *ok, void *stream) {
    if (!u0 || !status || !lo || !hi || !action || B < 0 || nu < 1) return MPCRL_E_ARG;
    if (B == 0) return 0;
    ON_DEVICE_OF(action);
    hipLaunchKernelGGL(policy_action_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, u0, (const int *)status, noise, lo, hi, B, nu, scale,
                       (float)sigma, (float)noise_clip, accept_status2, action, ok);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_replay_sample(const float *table, int row_len, int nx, int E, int cap, int steps, const int64_t *idx, int B, const int64_t *pos_t,
                        int exclude_pos, const uint8_t *iter_ok, float *rows, double *obs64, double *nxt64, int64_t *row_s, int32_t *cold_s,
                        int64_t *row_n, int32_t *cold_n, void *stream) {
    if (!table || !idx || !rows || !obs64 || !nxt64 || B < 1 || nx < 1 || row_len < 2 * nx + 2 || E < 1 || cap < 1 || steps < 1 || steps > cap) return MPCRL_E_ARG;
    if (iter_ok && (!pos_t || !row_s || !cold_s || !row_n || !cold_n)) return MPCRL_E_ARG;
    if (exclude_pos && (!pos_t || steps != cap || cap < 2)) return MPCRL_E_ARG;
    ON_DEVICE_OF(rows);
    ReplaySampleArgs a;
    a.table = table, a.row_len = row_len, a.nx = nx, a.B = B, a.E = E, a.cap = cap, a.steps = steps, a.idx = idx, a.pos_t = pos_t, a.exclude = exclude_pos, a.iter_ok = iter_ok;
    a.rows = rows, a.obs64 = obs64, a.nxt64 = nxt64, a.row_s = row_s, a.row_n = row_n, a.cold_s = cold_s, a.cold_n = cold_n;
    hipLaunchKernelGGL(replay_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_td3_cartpole_collect(const double *par, int E, double *state, int64_t *steps, const double *u0, const int32_t *status, const float *eps,
                               const double *u01, double lo, double hi, int scale, double sigma, double *obs, int32_t *ended, float *table, int cap,
                               double reward_scale, int64_t *pos, uint8_t *iter_ok, int64_t *iter_rows, double *stats, void *workspace, void *stream) {
    if (!par || E < 1 || !state || !steps || !u0 || !status || !eps || !u01 || !obs || !ended || !table || cap < 1 || !pos || !stats || !workspace)
        return MPCRL_E_ARG;
    ON_DEVICE_OF(state);
    Td3CollectArgs a;
    a.par.gravity = par[0], a.par.masscart = par[1], a.par.masspole = par[2], a.par.length = par[3], a.par.force_mag = par[4], a.par.tau = par[5];
    a.par.x_threshold = par[6], a.par.theta_threshold = par[7], a.par.max_episode_steps = (long)par[8];
    a.E = E, a.state = state, a.steps = steps, a.u0 = u0, a.status = (const int *)status, a.eps = eps, a.u01 = u01, a.lo = lo, a.hi = hi, a.scale = scale;
    a.sigma = (float)sigma, a.obs = obs, a.ended = ended, a.table = table, a.cap = cap, a.reward_scale = reward_scale, a.pos = pos, a.iter_ok = iter_ok;
    a.iter_rows = iter_rows, a.stats = stats, a.ticket = (unsigned int *)workspace, a.partial = (double *)((char *)workspace + 16);
    hipLaunchKernelGGL(td3_cartpole_collect_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_td3_policy_post(const double *msg, int n_theta, double lr, const double *mask, double tau, double *theta, double *theta_target,
                          double *step_out, const float *crit, float *crit_target, int n_crit, void *stream) {
    if (!msg || n_theta < 1 || !mask || !theta || !theta_target || !step_out || n_crit < 0 || (n_crit > 0 && (!crit || !crit_target))) return MPCRL_E_ARG;
    ON_DEVICE_OF(theta);
    const int n = n_theta > n_crit ? n_theta : n_crit;
    hipLaunchKernelGGL(td3_policy_post_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, msg, n_theta, lr, mask, tau, theta, theta_target,
                       step_out, crit, crit_target, n_crit);
    HIP_OK(hipGetLastError());
    return 0;
}

int64_t mpcrl_dpg_workspace_bytes(int B, int n_p) {
    if (B < 1 || n_p < 1) return MPCRL_E_ARG;
    return 16 + (int64_t)((B + DPG_ROWS - 1) / DPG_ROWS) * (n_p + 1) * (int64_t)sizeof(double);
}

int mpcrl_dpg_grad(const float *dq_da, const uint8_t *ok, const double *dpi_dp, int B, int nu, int n_p, const double *lo, const double *hi, int scale,
                   void *workspace, double *out, void *stream) {
    if (!dq_da || !dpi_dp || !workspace || !out || B < 1 || nu < 1 || nu > 8 || n_p < 1 || (scale && (!lo || !hi))) return MPCRL_E_ARG;
    ON_DEVICE_OF(out);
    DpgArgs a;
    a.dq_da = dq_da, a.ok = ok, a.dpi_dp = dpi_dp, a.B = B, a.nu = nu, a.n_p = n_p, a.scale = scale, a.lo = lo, a.hi = hi;
    a.ticket = (unsigned int *)workspace, a.partial = (double *)((char *)workspace + 16), a.out = out;
    hipLaunchKernelGGL(dpg_grad_kernel, dim3((B + DPG_ROWS - 1) / DPG_ROWS), dim3(128), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int64_t mpcrl_critic_workspace_bytes(int B, int nx, int nu, int n_critics) {
    if (B < 0 || nx < 1 || nu < 1 || nx + nu > CRITIC_DMAX || n_critics < 1 || n_critics > 2) return MPCRL_E_ARG;
    const int64_t n_blocks = (B + CRITIC_S - 1) / CRITIC_S, n_params = (int64_t)n_critics * (CRITIC_H * (nx + nu) + CRITIC_H * CRITIC_H + 3 * CRITIC_H + 1);
    return n_blocks * (n_params + 2) * (int64_t)sizeof(float);
}

int mpcrl_critic_td_grad(const float *rows, int row_stride, int B, int nx, int nu, const float *a_next, const uint8_t *ok_u, const float *params,
                         const float *params_target, int n_critics, double gamma, double out_scale, void *workspace, double *grad, float *loss_out,
                         uint8_t *ok_out, void *stream) {
    if (!rows || !a_next || !params || !params_target || !workspace || !grad || B < 1 || nx < 1 || nu < 1 || nx + nu > CRITIC_DMAX || n_critics < 1 ||
        n_critics > 2 || row_stride < 2 * nx + nu + 2)
        return MPCRL_E_ARG;
    ON_DEVICE_OF(grad);
    CriticArgs a;
    a.rows = rows, a.row_stride = row_stride, a.row_len = 2 * nx + nu + 2, a.B = B, a.nx = nx, a.nu = nu, a.n_critics = n_critics;
    a.a_next = a_next, a.ok_u = ok_u, a.params = params, a.params_target = params_target, a.gamma = (float)gamma;
    a.partial = (float *)workspace, a.ok_out = ok_out;
    const int n_blocks = (B + CRITIC_S - 1) / CRITIC_S, n_params = n_critics * (CRITIC_H * (nx + nu) + CRITIC_H * CRITIC_H + 3 * CRITIC_H + 1);
    if (nx + nu <= 16)
        hipLaunchKernelGGL(critic_td_partial_kernel<2>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(critic_td_partial_kernel<1>, dim3(n_blocks), dim3(128), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(critic_td_reduce_kernel, dim3((n_params + 1 + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float *)workspace, n_blocks, n_params,
                       nx + nu, out_scale, grad, loss_out);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_critic_dq_da(const float *obs, int obs_stride, int B, int nx, int nu, const float *act, const uint8_t *ok, const float *params, float *dq_da,
                       uint8_t *ok_out, void *stream) {
    if (!obs || !act || !params || !dq_da || B < 1 || nx < 1 || nu < 1 || nx + nu > CRITIC_DMAX || obs_stride < nx) return MPCRL_E_ARG;
    ON_DEVICE_OF(dq_da);
    CriticDqdaArgs a;
    a.obs = obs, a.obs_stride = obs_stride, a.B = B, a.nx = nx, a.nu = nu, a.act = act, a.ok = ok, a.params = params, a.dq_da = dq_da, a.ok_out = ok_out;
    hipLaunchKernelGGL(critic_dqda_kernel, dim3((B + 3) / 4), dim3(64), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_env_linear_step(const double *par, int B, double *state, const double *action, const double *u01, void *obs, int obs_f32,
                          double *cost, void *stream) {
    if (!par || B < 0 || !state || !action || !u01 || !cost) return MPCRL_E_ARG;
    if (B == 0) return 0;
    ON_DEVICE_OF(state);
    LinearEnvPar p;
    for (int i = 0; i < 4; ++i) p.A[i] = par[i];
    p.B[0] = par[4], p.B[1] = par[5], p.lb_noise = par[6], p.ub_noise = par[7];
    p.low[0] = par[8], p.low[1] = par[9], p.high[0] = par[10], p.high[1] = par[11];
    if (obs_f32)
        hipLaunchKernelGGL(env_linear_step_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, B, state, action, u01, (float *)obs, cost);
    else
        hipLaunchKernelGGL(env_linear_step_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, B, state, action, u01, (double *)obs, cost);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_env_cartpole_reset(int B, double *state, int64_t *steps, const uint8_t *mask, const double *u01, void *obs, int obs_f32,
                             void *stream) {
    if (B < 0 || !state || !steps || !u01) return MPCRL_E_ARG;
    if (B == 0) return 0;
    ON_DEVICE_OF(state);
    if (obs_f32)
        hipLaunchKernelGGL(env_cartpole_reset_kernel<float>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, state, steps, mask, u01, (float *)obs);
    else
        hipLaunchKernelGGL(env_cartpole_reset_kernel<double>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, state, steps, mask, u01, (double *)obs);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_get_iterate(mpcrl_handle h, double *x, double *u, double *pi, double *bnd, double *res, void *stream) {
    if (!h) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = h->B, N = h->N, nx = h->nx, nu = h->nu, nw = nx + nu;
    if (x) HIP_OK(hipMemcpyAsync(x, h->X, B * (N + 1) * nx * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (u) HIP_OK(hipMemcpyAsync(u, h->U, B * N * nu * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (pi) HIP_OK(hipMemcpyAsync(pi, h->PI, B * N * nx * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (bnd) HIP_OK(hipMemcpyAsync(bnd, h->BND, B * 10 * (N + 1) * nw * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (res) HIP_OK(hipMemcpyAsync(res, h->RES, B * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}

int mpcrl_get_lagrangian(mpcrl_handle h, double *L, void *stream) {
    if (!h || !L) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    HIP_OK(hipMemcpyAsync(L, h->LAG, (size_t)h->B * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int mpcrl_set_iterate(mpcrl_handle h, const double *x, const double *u, const double *pi, const double *bnd, void *stream) {
    if (!h || !x || !u || !pi) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t B = h->B, N = h->N, nx = h->nx, nu = h->nu, nw = nx + nu;
    HIP_OK(hipMemcpyAsync(h->X, x, B * (N + 1) * nx * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(h->U, u, B * N * nu * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(h->PI, pi, B * N * nx * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (bnd)
        HIP_OK(hipMemcpyAsync(h->BND, bnd, B * 10 * (N + 1) * nw * sizeof(double), hipMemcpyDeviceToDevice, st));
    else {   // multipliers 0, slacks t = 1: the state MPCRL_COLD would build
        int rc = fill_cold_iterate(h, nullptr, false, st);
        if (rc) return rc;
    }
    h->have_iterate = true, h->dual_cold = bnd == nullptr;
    return 0;
}

int mpcrl_get_iterate_rows(mpcrl_handle h, double *x, double *u, double *pi, double *bnd, const int64_t *index, void *stream) {
    if (!h || !(x || u || pi || bnd)) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    const int N = h->N, nx = h->nx, nu = h->nu, nw = nx + nu;
    hipLaunchKernelGGL(iterate_rows_kernel<false>, dim3((unsigned)h->B), dim3(256), 0, (hipStream_t)stream, h->X, h->U, h->PI, h->BND, x, u, pi, bnd,
                       (const long *)index, (N + 1) * nx, N * nu, N * nx, 10 * (N + 1) * nw);
    HIP_OK(hipGetLastError());
    return 0;
}

int mpcrl_set_iterate_rows(mpcrl_handle h, const double *x, const double *u, const double *pi, const double *bnd, const int64_t *index, void *stream) {
    if (!h || !x || !u || !pi) return MPCRL_E_ARG;
    ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int N = h->N, nx = h->nx, nu = h->nu, nw = nx + nu;
    if (!bnd) {   // multipliers 0, slacks t = 1: the state MPCRL_COLD would build
        int rc = fill_cold_iterate(h, nullptr, false, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(iterate_rows_kernel<true>, dim3((unsigned)h->B), dim3(256), 0, st, h->X, h->U, h->PI, h->BND, (double *)x, (double *)u, (double *)pi,
                       (double *)bnd, (const long *)index, (N + 1) * nx, N * nu, N * nx, 10 * (N + 1) * nw);
    HIP_OK(hipGetLastError());
    h->have_iterate = true, h->dual_cold = bnd == nullptr;
    return 0;
}

}  // extern "C"
