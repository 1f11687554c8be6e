// mpcrl_host.hpp — what the translation units of libmpcrl_hip.so share on the host side: the handle, the error macro, and the table
// through which mpcrl_api.hip reaches the chain-of-masses kernels.  The library is built from one translation unit per chain size
// (chain_inst.hip, -DMPCRL_CHAIN_NMASS=3..7) next to mpcrl_api.hip (C ABI, cartpole / linear-system kernels, library kernels), so that
// the sizes compile in parallel (csrc/Makefile); every unit carries its own device code, nothing is linked on the device side.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/mpcrl.h"
#include "chain_common.hpp"

using mpcrl::LargeArgs;
using mpcrl::LargeSpec;
using mpcrl::SmallSpec;

struct MpcrlSolver {
    int model, B, device, nx, nu, np, N;
    SmallSpec small;
    LargeSpec large;
    bool is_large = false;
    int n_mass = 0;
    double *ws = nullptr, *consts_dev = nullptr;
    int *perm = nullptr, *cold_mask = nullptr;
    bool have_perm = false, have_cold_mask = false;
    double *order_state = nullptr;   // order_kernel.hpp: {spread, minimum, coordinate} of the last from-scratch packing order
    unsigned order_calls = 0;
    size_t ws_stride = 0;
    double *theta = nullptr;   // [np] or [B, np]
    int theta_stride = 0;
    double *X = nullptr, *U = nullptr, *PI = nullptr, *BND = nullptr, *RES = nullptr, *LAG = nullptr;
    int64_t bytes = 0;
    int n_simd = 1024;          // SIMDs of the device (one resident wavefront each for the small solve kernel)
    int slice_mode = 0;         // mpcrl_set_launch_mode / MPCRL_TIME_SLICE: 0 = automatic, 1 = whenever legal, -1 = never
    int linear_spl = 3;         // linear-system model: stages per lane of the solve kernel (3: linear_kernel.hpp; 1: small_solve_kernel)
    // automatic mode: the two launch shapes of the small solve kernel are timed against each other on the caller's own batches
    // (choose_launch below): [0] = time-sliced, [1] = plain
    struct Tuner {
        hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
        bool pending[2] = {false, false};
        float ms[2] = {-1.f, -1.f};
        unsigned calls = 0;
    } tune[2];                  // [0] cold calls, [1] warm calls (different work per instance: timed separately)
    int planned = -1;           // what mpcrl_query_time_sliced promised for the next solve (-1: nothing promised)
    bool have_iterate = false;
    bool dual_cold = false;   // the stored bound multipliers are placeholders (set_iterate without bnd): next solve = MPCRL_COLD_DUAL
};

#define HIP_OK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "mpcrl: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MPCRL_E_HIP;                                                               \
        }                                                                                     \
    } while (0)

// One chain size as mpcrl_api.hip sees it (defined by chain_inst.hip, one per -DMPCRL_CHAIN_NMASS)
struct MpcrlChainEntry {
    int n_mass;
    long np;                                                        // length of the parameter vector (ChainDev<n>::NP)
    size_t (*ws_doubles)(int N);                                    // per-instance workspace (LargeLayout)
    bool (*fits)(int N);                                            // every dynamic-LDS request of the kernels fits one workgroup
    int (*launch)(MpcrlSolver *h, LargeArgs a, hipStream_t st);     // init + SQP (+ sensitivity) launches of one mpcrl_solve
    int (*debug_phases)(unsigned long long *out16, int reset);      // -DMPCRL_PROFILE_PHASES builds: the unit's phase ticks (else null)
};
const MpcrlChainEntry *mpcrl_chain_entry_3(), *mpcrl_chain_entry_4(), *mpcrl_chain_entry_5(), *mpcrl_chain_entry_6(), *mpcrl_chain_entry_7();
