// chain_inst.hip — the chain-of-masses kernels of ONE chain size (-DMPCRL_CHAIN_NMASS=3..7) and their launch sequence; one translation
// unit per size so that the sizes compile in parallel (mpcrl_host.hpp, csrc/Makefile).
#include "mpcrl_host.hpp"

#include "chain_kernel.hpp"

#ifndef MPCRL_CHAIN_NMASS
#error "compile with -DMPCRL_CHAIN_NMASS=3..7"
#endif

using namespace mpcrl;

namespace {

using Model = ChainDev<MPCRL_CHAIN_NMASS>;

// dynamic LDS of the chain kernels for a horizon of N stages (bytes): the SQP / adjoint-Riccati kernels, the second-order point pass,
// the mixed-term kernel, the output reduction.  mpcrl_create refuses a (chain size, horizon) pair whose largest request does not fit.
template <class M>
struct LargeLds {
    unsigned sqp, point, mix2, out;
    explicit LargeLds(int N)
        : sqp((unsigned)(ChainCfg<M>::lds_doubles(N) * sizeof(double))),
          point((unsigned)((M::NTD + N * M::NX) * sizeof(double))),
          mix2((unsigned)((M::NTD + M::NX) * 64 * sizeof(double))),
          out((unsigned)((64 + (1 + M::NU) * ((N + 1) * M::NX + N * M::NU)) * sizeof(double))) {}
    bool fits() const {   // 64 KB per workgroup; the SQP kernel also holds ~1 KB of static LDS
        return sqp + 1024 <= 64 * 1024 && point <= 64 * 1024 && mix2 <= 64 * 1024 && out <= 64 * 1024;
    }
};

template <class M>
int launch_large(MpcrlSolver *h, LargeArgs a, hipStream_t st) {
    a.ws = h->ws, a.ws_stride = h->ws_stride;
    const int B = h->B, N = h->N;
    const LargeLds<M> lds(N);
    if (!lds.fits()) return MPCRL_E_ARG;   // (mpcrl_create has checked: not reached)
    hipLaunchKernelGGL(chain_init_kernel<M>, dim3(B), dim3(64), 0, st, h->large, a);
    // the whole SQP loop of an instance runs inside one wavefront of one launch (linearisation, QP, step; chain_linearise.hpp)
    hipLaunchKernelGGL(chain_sqp_kernel<M>, dim3(B), dim3(64), lds.sqp, st, h->large, a);
    HIP_OK(hipGetLastError());
    if (a.flags & (MPCRL_SENS_V | MPCRL_SENS_PI)) {
        const bool want_pi = (a.flags & MPCRL_SENS_PI) && a.dpi && !a.u0fix;
        // second-order point pass, then grad_theta (nu' F) from its tables: the same evaluation order for dV/dp whatever the flags
        hipLaunchKernelGGL((chain_point_kernel<M, true>), dim3((unsigned)B), dim3(64), lds.point, st, h->large, a);
        hipLaunchKernelGGL(chain_sens_th2_kernel<M>, dim3((unsigned)(((long)B * N + 63) / 64)), dim3(64), 0, st, h->large, a);
        if (want_pi) {
            hipLaunchKernelGGL(chain_sens_ad_kernel<M>, dim3((unsigned)(B * HexCfg<M>::groups(N))), dim3(64), 0, st, h->large, a);
            hipLaunchKernelGGL(chain_sens_riccati_kernel<M>, dim3(B), dim3(64), lds.sqp, st, h->large, a);
            hipLaunchKernelGGL(chain_sens_mix2_kernel<M>, dim3((unsigned)(((long)B * N * M::NU + 63) / 64)), dim3(64), lds.mix2, st, h->large, a);
        }
        // one workgroup of 1024 lanes per instance; its trajectories staged in LDS (chain_sens_out_kernel)
        hipLaunchKernelGGL(chain_sens_out_kernel<M>, dim3((unsigned)B), dim3(1024), lds.out, st, h->large, a);
        HIP_OK(hipGetLastError());
    }
    return 0;
}

size_t ws_doubles(int N) { return LargeLayout<Model>(N).total; }
bool lds_fits(int N) { return LargeLds<Model>(N).fits(); }

#ifdef MPCRL_PROFILE_PHASES
int debug_phases(unsigned long long *out, int reset) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {0};
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof(z)));
    }
    return 0;
}
#else
constexpr int (*debug_phases)(unsigned long long *, int) = nullptr;
#endif

}  // namespace

#define MPCRL_CAT_(a, b) a##b
#define MPCRL_CAT(a, b) MPCRL_CAT_(a, b)
const MpcrlChainEntry *MPCRL_CAT(mpcrl_chain_entry_, MPCRL_CHAIN_NMASS)() {
    static const MpcrlChainEntry e = {MPCRL_CHAIN_NMASS, (long)Model::NP, ws_doubles, lds_fits, launch_large<Model>, debug_phases};
    return &e;
}
