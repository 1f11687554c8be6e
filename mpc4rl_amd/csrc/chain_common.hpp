// chain_common.hpp — data layout and wavefront-level building blocks of the chain-of-masses solver (chain_kernel.hpp).
//
// The chain kernels replace, for a whole batch at once, what the reference does per instance through
//   ocp_solver.solve()            rlmpc/mpc/common/mpc.py:42,79,195     (acados SQP + HPIPM, not vendored)
//   update_nlp(): dL_dp, dpi_dp   rlmpc/mpc/nlp.py:1399-1424            (dense 2385 x 2385 Jacobian + SuperLU, 499 right-hand sides)
// for OCPs whose stage blocks do not fit one lane (chain of masses: nx = 9 .. 33, nu = 3, N = 40;
// rlmpc/mpc/chain_mass/ocp_utils.py:59-147,195-316).
//
// Mapping on gfx950.  ONE WAVEFRONT PER OCP INSTANCE, one wavefront per SIMD (a batch of 1024 instances is exactly one wavefront on each
// of the chip's 1024 SIMDs); no workgroup barrier anywhere: the lanes of a wavefront exchange data through LDS or through the
// instance's HBM workspace, and because the memory operations of a wavefront are performed in order a wavefront-scope fence (a
// compiler barrier, no s_waitcnt) is all the ordering needed.
//   * chain_sweeps.hpp     — the QP: Mehrotra predictor-corrector whose three Riccati sweeps (factor, backward vector, forward) are
//                            register-resident v_mfma_f64_16x16x4 pipelines in the "Omega" coordinates of OmCfg below.  No LDS in the
//                            sweeps: operands stream from HBM in their register layout (the closed-loop blocks G_k are written once
//                            and read three times per interior-point iteration; a horizon of them does not fit on-chip),
//                            the vectors of the horizon are staged in LDS in Omega order, the bound rows live in registers.
//   * chain_linearise.hpp  — the SQP loop of an instance inside ONE launch (chain_sqp_kernel): per round the linearisation of the
//                            2-step RK4 map (point pass: one lane per stage; direction pass: one lane per (stage, direction)), cost
//                            and residuals, the QP, the full step.  The big phases are real calls with register allocations of
//                            their own (fused by inlining, the jets of one phase pushed the operands of another into scratch, and
//                            every scratch reload is an s_waitcnt vmcnt(0) that also drains the streaming loads).
//   * chain_sens.hpp       — dV/dp and du0*/dp: exact Lagrangian Hessian per stage, ONE adjoint Riccati factorisation + one forward
//                            sweep for the nu adjoint solves, the mixed term on the point tables, the output reductions.
// Only hard box bounds are supported here (the chain problem has bounds on u only).
//
// The iteration is the one of small_kernel.hpp / DESIGN.md §2 (same constants), so results agree with the oracle to rounding.
#pragma once
#include <type_traits>

#include "small_kernel.hpp"

namespace mpcrl {

constexpr int LARGE_MAXNW = 40;
// complementarity tolerance of an inexact QP = this x its residual tolerance (0.1 in the small solvers): with the cap itself the
// n_mass 7 chain needs 15.8 instead of 18.8 interior-point iterations per solve; n_mass 3 / 5 and every SQP iteration count unchanged
constexpr double CHAIN_TOL_MU_FACTOR = 1.0;

#ifndef MPCRL_CHAIN_MERGE_CALLS
#ifdef MPCRL_PROFILE_PHASES
#define MPCRL_CHAIN_MERGE_CALLS 0   // (the phase profile times the sweeps one by one)
#else
#define MPCRL_CHAIN_MERGE_CALLS 1   // predictor = factor + forward, corrector = backward + forward as ONE phase call each (half the callee-saved register traffic)
#endif
#endif

struct LargeSpec {
    int N, np, cost_kind, rk_steps, max_iter;
    int exit_window;        // opt-in divergence exit (mpcrl_set_exit_rule), as in SmallSpec: 0 = off
    double exit_factor;
    double dT, gamma, h, tol;
    double lb0[4], ub0[4];
    double lb[LARGE_MAXNW], ub[LARGE_MAXNW], lbe[LARGE_MAXNW], ube[LARGE_MAXNW];
    const double *consts;   // device: x_ss
};

struct LargeArgs {
    int B, flags, theta_stride;
    const int *perm;
    const int *cold;                  // [B] or null: per-instance MPCRL_COLD
    const double *x0, *u0fix, *theta;
    double *X, *U, *PI, *BND, *RES;   // iterate (layouts of mpcrl_get_iterate)
    double *LAG;                      // [B] Lagrangian of the mirror at the returned iterate (mpcrl_get_lagrangian)
    double *ws;                       // per-instance workspace, ws_stride doubles each
    size_t ws_stride;
    double *u0_out, *V, *dV, *dpi;
    int *status, *iters;
};

// ---- Omega coordinates of the register-resident sweeps (round 4).
// v_mfma_f64_16x16x4 wants A(i, k) and B(k, j) at lane 16 k + (i | j) and returns register r = rows 4 r + lane / 16, column lane % 16:
// the RESULT layout of a matrix (row group of 4 on lane / 16, column on lane % 16) is at the same time its layout as a B operand
// (contraction over its rows) and, transposed, as an A operand — so a chain of products can stay in registers as long as every
// matrix of the recursion is indexed by ONE index set on both sides.  That set is Omega = 0 .. NW - 1, NW = NX + NU slots:
//     slot e <  Q           : state x_e                    Q = 4 floor(NX / 4)
//     slot Q + l, l < NU    : control u_l  as a COLUMN index (stage vector), a zero pad row as a ROW index (next state)
//     slot e >= Q + NU      : state x_{e - NU}
// i.e. the stage vector [x; u] with the controls moved into one aligned group of four slots (one register, lanes lr = 0 .. NU - 1),
// and column NW (= VC) next to it carries the VECTORS of the recursion through the same products:
//     W  = [A B | b]   (rows: next state, pad rows 0)       T = P W = [P A, P B | P b (+ p)]
//     M  = H + D + W' T   ->  column VC = g + W'(P b + p),  rows Q.. = [S | R | mv_u]
//     K  = R^-1 [S | R | mv_u] (one MFMA per column tile, R^-1 from a Cholesky every lane runs on broadcast values)
//     P' = M - S' K    ->  x / x block = P_k, column VC = p_k          G = W - B K with K written into the pad rows
// Per stage the factor sweep streams G_k (closed loop: x rows [Acl | B], pad rows [-K | 0]) and P_k to HBM in this register
// layout (64 lanes x 8 bytes per register: every access a full 512-byte burst), and the two vector sweeps are MFMA chains whose
// operand vector IS the previous stage's result registers:
//     backward  [p_k; mv_u] = g~ + G' [p_{k+1} + P_{k+1} b; g~_u]          forward  [dx_{k+1}; du_k] = [b; -kff] + G [dx_k; -kff]
template <class M>
struct OmCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU;
    static constexpr int Q = 4 * (NX / 4), VC = NW, RG = (NW + 3) / 4, NT = (NW + 16) / 16, NTR = (RG + 3) / 4;
    static constexpr int GQ = Q / 4, TQ = Q / 16, RQ = GQ % 4, LQ = Q % 16, TV = VC / 16, LV = VC % 16;
    // streamed stage block: per row group the full column tiles (64 lanes each), then the columns < NW of the last tile compactly
    // (4 x LV doubles, lane lr LV + lc): exactly the NW x NW entries the vector sweeps read when NW is a multiple of 4
    static constexpr int CT = 4 * LV, GSZ = RG * (TV * 64 + CT);
    // stride of the per-stage Omega vectors: 4 RG slots + 2, so that the 16 stages a batched MFMA pass reads as its 16 operand columns
    // (lane lc -> stage k0 + lc) fall on 16 different bank pairs (4 RG doubles alone: a 4-way conflict at n_mass 5, 2-way at n_mass 7)
    static constexpr int HBS = 4 * RG + 2;
    MPCRL_DI static unsigned goff(int rg, int tj, int lr, int lc) {   // register (rg, tj) of lane (lr, lc) inside a block
        if (tj < TV) return (unsigned)((rg * TV + tj) * 64 + lr * 16 + lc);
        return (unsigned)(RG * TV * 64 + rg * CT + lr * LV + (lc < LV ? lc : (LV > 0 ? LV - 1 : 0)));
    }
    static_assert(NU <= 4 && LQ + NU <= 16 && NTR <= NT, "the control group sits in one register, inside one column tile");
    // slot -> index in the stage vector [u; x] (e < NW) / index of the state (-1: none)
    MPCRL_DI static constexpr int nat(int e) { return e < Q ? NU + e : (e < Q + NU ? e - Q : e); }
    MPCRL_DI static constexpr int xrow(int e) { return e < Q ? e : (e < Q + NU ? -1 : (e < NW ? e - NU : -1)); }
};

// per-instance workspace layout (doubles)
template <class M>
struct LargeLayout {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD;
    // stage block of the linearisation: rows 0 .. NX - 1 = [B A]_k (columns in stage-vector order [u; x]), rows NX .. NX + NU - 1 =
    // [0 | -K_k], the feedback gain the factor sweep of the current interior-point iteration left there — one block, one address
    // pattern for the vector sweeps, which multiply with [A B] and K instead of a stored closed-loop matrix
    static constexpr int BAS = NW * NW;
    size_t BA, r, q, dx, du, nuq, Dx, Du, rg, rb, rt, Dg, lamw, tw, aff, p, kff, Hex, term, ynu, state, Ydx, Ydu, Ydnu,
        term2, ptab, gtab, qvtab, G2, P2, hb2, minv2, mvu2, total;
    __host__ __device__ explicit LargeLayout(int N) {
        size_t o = 0;
        auto take = [&](size_t n) { size_t s = o; o += (n + 1) & ~(size_t)1; return s; };   // every array 16-byte aligned
        BA = take((size_t)N * BAS + 2), r = take((size_t)N * NX), q = take((size_t)(N + 1) * NW);
        // (+ 2: a dump slot behind the array — masked lanes of the sweeps store there instead of branching)
        dx = take((size_t)(N + 1) * NX), du = take((size_t)N * NU), nuq = take((size_t)(N + 1) * NX + 2);
        Dx = take((size_t)(N + 1) * NX + 2), Du = take((size_t)N * NU + 2);
        rg = take((size_t)(N + 1) * NW), rb = take((size_t)N * NX), rt = take((size_t)(N + 1) * NW), Dg = take((size_t)(N + 1) * NW);
        lamw = take((size_t)2 * (N + 1) * NW), tw = take((size_t)2 * (N + 1) * NW), aff = take((size_t)2 * (N + 1) * NW);
        p = take((size_t)(N + 1) * NX + 2), kff = take((size_t)N * NU + 2);
        Hex = take((size_t)(N + 1) * NW * NW), term = take((size_t)N * NTD), ynu = take((size_t)(N + 1) * NX);
        state = take(16);   // ST_* below: the SQP loop's per-instance state between launches
        Ydx = take((size_t)NU * (N + 1) * NX + 2), Ydu = take((size_t)NU * N * NU), Ydnu = take((size_t)NU * (N + 1) * NX + 2);   // adjoint solutions
        term2 = take((size_t)NU * N * NTD);
        // per stage: coefficients of the 8 evaluation points of the RK4 map (chain_point_kernel), and the link Hessians of the adjoint
        ptab = take((size_t)N * 8 * M::NL * M::TAB2), gtab = take((size_t)N * 8 * M::NL * 6);
        qvtab = take((size_t)N * 8 * M::NL * 6);      // per evaluation point and link: force adjoint q (3), velocity difference dv (3) — chain_sens_mix2
        // the sweeps' streams: closed-loop blocks G_k, cost-to-go P_k of the adjoint factorisation (Omega register layout), hb_k = P_{k+1} b_k, R_k^-1, mv_u of the corrector
        G2 = take((size_t)N * OmCfg<M>::GSZ), P2 = take((size_t)(N + 1) * OmCfg<M>::GSZ + 64);
        hb2 = take((size_t)N * OmCfg<M>::HBS), minv2 = take((size_t)N * 16), mvu2 = take((size_t)N * 4);
        total = (o + 7) & ~(size_t)7;
    }
};

enum { ST_ACTIVE = 0, ST_IT = 1, ST_NIPM = 2, ST_TIGHT = 3, ST_STEPN = 4, ST_COST = 5, ST_RES = 6, ST_STATUS = 10 };

template <class M, bool SECOND, bool TH_LDS>
__device__ __forceinline__ void chain_point_body(const double *X, const double *U, const double *th, double *w, int N, int k, double h, int steps, double *lacc);
template <class M>
__device__ __forceinline__ void chain_dir_body(const double *th, double *w, double *tabl, int N, int lane, double h, int steps);
template <class M, bool SECOND, bool TH_LDS = false>
__device__ void chain_point_pass(const double *X, const double *U, const double *th, double *w, int N, int k, double h, int steps, double *lacc = nullptr);

// The phase functions of the chain solver are real calls (register allocations of their own; the scratch they report is the
// save / restore of callee-saved registers in their prologue and epilogue, not traffic inside their loops).
#define MPCRL_PHASE_FN __attribute__((noinline))

// An array inside the instance's workspace: one base pointer for all of them (scalar registers) plus a 32-bit offset, so that
// every access is `global_load/store v, voffset, s[base]` — no 64-bit per-lane address arithmetic to keep live.
struct WsArr {
    char *base;
    unsigned off;   // doubles
    MPCRL_DI double &operator[](int i) const { return *(double *)(base + ((off + (unsigned)i) << 3)); }
    MPCRL_DI WsArr operator+(int i) const { return WsArr{base, off + (unsigned)i}; }
    MPCRL_DI explicit operator bool() const { return base != nullptr; }
};

// Development aid: -DMPCRL_PROFILE_PHASES accumulates shader-clock ticks of lane 0 per phase, summed over the wavefronts
// (read through mpcrl_debug_phases; profiles/microbench/chain_phases.py).  Off in the product build.

// A pointer that arrives through a real call has lost its address space: every access through it is a FLAT instruction, which counts
// on lgkmcnt as well as vmcnt — a wait for an LDS read then also waits for every global load in flight (the prefetches), and the
// compiler can no longer order the two streams.  These give it back (the round trip through the address-space-qualified type is what
// InferAddressSpaces follows).
template <class T>
MPCRL_DI T *as_global(T *ptr) {
    typedef __attribute__((address_space(1))) T GT;
    return (T *)(GT *)(unsigned long long)ptr;
}
template <class T>
MPCRL_DI T *as_lds(T *ptr) {
    typedef __attribute__((address_space(3))) T LT;
    return (T *)(LT *)(unsigned long)(unsigned)(unsigned long long)ptr;
}

// Ordering between the lanes of ONE wavefront (LDS and global alike): memory operations of a wavefront are performed in order,
// so only the compiler has to be kept from moving accesses across this point — no s_waitcnt is emitted.
MPCRL_DI void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
MPCRL_DI double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}
MPCRL_DI double wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s));
    return v;
}

typedef double d2_t __attribute__((ext_vector_type(2)));

// A lone wavefront per SIMD has nothing to switch to while an LDS read is in flight (~64-130 cycles), and left to itself the
// scheduler interleaves every read with its use.  The hot loops therefore stage a whole batch of operands into registers,
// fence the scheduler, and only then start the arithmetic: one exposed LDS latency per batch instead of one per operand.
#define MPCRL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Opaque copy of a per-lane value.  Every phase of the solver starts from a laundered lane index: whatever it derives from it
// (addresses, predicates, tile origins) then cannot be hoisted out of the interior-point loop, where the optimiser would
// otherwise keep hundreds of such loop invariants live across all phases and spill them (each reload is an s_waitcnt vmcnt(0)).
MPCRL_DI int launder(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// dot product of NN LDS operands (stride rs) with an LDS vector, in chunks of at most 12 (a double is two registers and only
// 256 of the 512 are directly usable by the vector ALU: the batches have to stay small); four partial sums
template <int NN>
MPCRL_DI double lds_dot(const double *row, int rs, const double *vec, double init) {
    constexpr int CH = NN <= 12 ? NN : (NN % 12 == 0 ? 12 : (NN % 11 == 0 ? 11 : (NN % 8 == 0 ? 8 : (NN % 7 == 0 ? 7 : 3))));
    static_assert(NN % CH == 0, "chunking");
    double acc[4] = {init, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < NN / CH; ++c) {
        double a[CH], b[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) a[j] = row[(c * CH + j) * rs], b[j] = vec[c * CH + j];
        MPCRL_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j & 3] = fma(a[j], b[j], acc[j & 3]);
        MPCRL_SCHED_FENCE();
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int D, int d = 0, class F>
MPCRL_DI void static_for(F &&f) {
    if constexpr (d < D) {
        f(std::integral_constant<int, d>{});
        static_for<D, d + 1>(f);
    }
}
// Software pipeline over the stages of a sweep.  The operands of D stages are in flight in registers (slot = compile-time index,
// so the register arrays are never runtime-indexed); body(idx, slot, refill) consumes slot `slot` for stage number idx (in
// processing order) and calls refill() once the slot's registers are free, which issues the loads of stage idx + D.  A single
// wavefront has nobody to switch to while a load is outstanding: the depth is what hides the HBM latency (~1-2 us under load).
// The steady state is straight-line code: refills past the end re-request the last stage instead of branching, and every block
// load / LDS store below is unconditional (the arrays are padded), so that s_waitcnt vmcnt(n) can be exact — behind a branch
// the wait-count analysis falls back to vmcnt(0), which waits for the loads just issued and makes the depth useless.
template <int D, class Fetch, class Body>
MPCRL_DI void staged_loop(int n, Fetch &&fetch, Body &&body) {
    static_for<D>([&](auto s) { fetch(s.value < n ? s.value : n - 1, s); });
    const int full = n - n % D;
    for (int base = 0; base < full; base += D)
        static_for<D>([&](auto s) {
            const int idx = base + s.value;
            body(idx, s, [&] { fetch(idx + D < n ? idx + D : n - 1, s); });
        });
    static_for<D>([&](auto s) {
        const int idx = full + s.value;
        if (idx < n) body(idx, s, [&] {});
    });
}

// Element-wise pass over n workspace entries, lane-strided, in batches of CH: the loads of a whole batch are issued before any of
// its results is stored.  Written as a plain `for (e = lane; e < n; e += 64)` such a pass is one global-memory round trip (1-2 us)
// per iteration: the arrays of the workspace hang off one base pointer, so every store may alias the next load and the compiler
// keeps them in program order.  load(e) returns the operands of entry e (entries past n re-read the last one), body(e, v) stores.
template <int CH, class Load, class Body>
MPCRL_DI void batched_pass(int n, int lane, Load &&load, Body &&body) {
    for (int base = lane; base < n; base += 64 * CH) {
        decltype(load(0)) v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = base + 64 * c;
            v[c] = load(e < n ? e : n - 1);
        }
        MPCRL_SCHED_FENCE();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = base + 64 * c;
            if (e < n) body(e, v[c]);
        }
    }
}
struct Pair2 {
    double a, b;
};
struct Quad4 {
    double a, b, c, d;
};

// LDS plan of the kernels that run the solver (chain_sqp_kernel, chain_sens_riccati_kernel) and the few compile-time choices that
// depend on the chain size
template <class M>
struct DirCfg;
template <class M>
struct ChainCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU;
    static constexpr int NBA2 = (NX * NW / 2 + 63) / 64;    // 16-byte pieces per lane of one [B A] block
    static constexpr int DEPTH = NX <= 21 ? 3 : 2;          // [B A] blocks in flight in round_start's stage pass (n_mass 6, 7)
    // LDS (doubles): the small vectors that live for the whole kernel first, then the big region the phases share
    static constexpr int ev(int n) { return n + (n & 1); }
    static constexpr int oBB = 0, oCK = oBB + ev(NX), oLB = oCK + 64, oBig = oLB + 4 * NW + 8;
    // start of an SQP round: Q, then X - x_ss and U of the whole horizon (up to 64 stages); its stage pass (n_mass 6, 7) publishes
    // [B A]_k where X sat
    static constexpr int oQ = oBig, oX = oBig + NW * NW, oU = oX + 64 * NX, oBA = oBig + NW * NW;
    static constexpr int BIG_R = NW * NW + 64 * NW;
    // [B A]_k' nu_{k+1} of the stationarity residual out of the direction pass.  Up to n_mass 5: at n_mass 7 the lane's four tangent
    // arrays already overflow the vector registers and the extra dot product costs the direction pass more (+450 us per solve) than
    // the stage pass it replaces (-380 us).
    static constexpr bool FUSE_GT = NX <= 21;
    // The sweeps stage the Hessian table of the factor sweep, or two vectors of the whole horizon in Omega order (HBS doubles per
    // stage), in the big region — which depends on the horizon, so the kernels take their LDS as a launch argument.
    __host__ __device__ static constexpr int lds_doubles(int N) {
        int big = BIG_R;
        const int tab = OmCfg<M>::RG * OmCfg<M>::NT * 64, vec = 2 * (N + 1) * OmCfg<M>::HBS, cst = (N + 1) * OmCfg<M>::HBS + tab;
        big = big > tab ? big : tab, big = big > vec ? big : vec, big = big > cst ? big : cst;
        // the direction pass: its tables, the compact parameter copy and the multipliers of the whole horizon
        // (and the point pass's RK4 accumulators: NX per stage lane)
        const int pacc = N * NX > 64 * DirCfg<M>::ACCL ? N * NX : 64 * DirCfg<M>::ACCL;      // (the direction pass's accumulators reuse the region)
        const int dir = DirCfg<M>::CO + M::NTD + (FUSE_GT ? (N + 1) * NX : 0) + pacc;
        big = big > dir ? big : dir;
        return oBig + big + (big & 1);
    }
    static_assert((oBig + BIG_R) * 8 <= 40 * 1024, "four wavefronts per CU");
};

}  // namespace mpcrl
