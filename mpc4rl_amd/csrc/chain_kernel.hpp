// chain_kernel.hpp — SQP / Riccati-IPM / adjoint-sensitivity kernels for OCPs whose stage blocks do not fit one lane
// (chain of masses: nx = 9 / 21 / 33, nu = 3, N = 40; rlmpc/mpc/chain_mass/ocp_utils.py:59-147,195-316).
//
// Replaces, for a whole batch at once, what the reference does per instance through
//   ocp_solver.solve()            rlmpc/mpc/common/mpc.py:42,79,195     (acados SQP + HPIPM, not vendored)
//   update_nlp(): dL_dp, dpi_dp   rlmpc/mpc/nlp.py:1399-1424            (dense 2385 x 2385 Jacobian + SuperLU, 499 right-hand sides)
//
// Mapping on gfx950 (MI355X-first).  The work has two shapes and each gets the launch geometry that suits it:
//   * DERIVATIVES of the 2-step RK4 map for the sensitivities (Hessian columns, parameter gradients) are independent per (instance,
//     stage, direction): grid-wide kernels with one item per lane, everything in registers (chain_sens_ad_kernel,
//     chain_sens_mix_kernel).  They want ~350-500 registers per lane.  The Jacobians [B A]_k of the SQP rounds are computed by the
//     instance's own wavefront (chain_point_pass / chain_dir_pass, non-inlined calls with register allocations of their own).
//   * the RICCATI interior-point solve of one QP is a dependency chain over the stages: ONE WAVEFRONT PER OCP INSTANCE, one
//     wavefront per SIMD (a batch of 1024 instances is exactly one wavefront on each of the chip's 1024 SIMDs).  There is no
//     workgroup barrier anywhere: lanes of one wavefront exchange data through LDS or through the instance's HBM workspace, and
//     because the memory operations of a wavefront are performed in order a wavefront-scope fence (a compiler barrier, no
//     s_waitcnt) is all the ordering needed.  Global stores are fire-and-forget.
//       - factor sweep: P_{k+1}, [B A]_k, T = P [B A] and M = H + D + [B A]' T of the CURRENT stage live in LDS; the two stage GEMMs
//         are register-tiled (TI x TJ / TS x TS outputs per lane, operands read as LDS vectors shared by the tile), the operands of
//         stage k-1 are fetched from HBM while stage k is computed.  Per-stage results (P_k, K_k, L_k and the CLOSED-LOOP matrix
//         Acl_k = A_k - B_k K_k) stream to HBM: a horizon of factors does not fit on-chip.
//       - the vector sweeps of the corrector / the extra right-hand sides are pure matrix-vector chains on Acl_k
//             p_k = (g_x - K' g_u) + Acl_k' (p_{k+1} + P_{k+1} b_k),        dx_{k+1} = Acl_k dx_k + (b_k - B_k kff_k)
//         with everything that does not sit on the chain (P_{k+1} b_k, kff_k, du_k, the multiplier step) done stage-parallel.
//   * the SQP loop of an instance runs inside ONE launch (chain_sqp_kernel): per round linearisation, cost / residuals / stopping
//     test, QP, full step.  What keeps the Riccati loops free of spills next to the derivative code is the call boundary: fused by
//     inlining, the jets of the derivative code pushed the loops' operands into scratch, and every scratch reload is an
//     s_waitcnt vmcnt(0) that also drains the streaming stores.
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
#ifndef MPCRL_CHAIN_SCALE_RES
#define MPCRL_CHAIN_SCALE_RES 1
#endif

// Round 4: the three Riccati sweeps (factor, backward vector sweep, forward sweep) as register-resident MFMA pipelines in the
// "Omega" coordinates of OmCfg below (no LDS in the sweeps, operands straight from HBM in their register layout).  0 = the round-3
// sweeps (operands staged through LDS), kept for A/B measurements.
#ifndef MPCRL_CHAIN_V2
#define MPCRL_CHAIN_V2 1
#endif
#ifndef MPCRL_CHAIN_V2_MAXNX
#define MPCRL_CHAIN_V2_MAXNX 33   // largest state dimension that runs the round-4 sweeps
#endif
#ifndef MPCRL_CHAIN_MIX2
#define MPCRL_CHAIN_MIX2 1     // chain_sens_mix2_kernel (point tables) instead of chain_sens_mix_kernel (jets)
#endif
#ifndef MPCRL_CHAIN_ROWS_IN_REGS
#define MPCRL_CHAIN_ROWS_IN_REGS 1     // qp_solve_rows: the bound rows of a QP in registers (at most 128 rows)
#endif
#ifndef MPCRL_CHAIN_TH2
#define MPCRL_CHAIN_TH2 1       // with du0*/dp: grad_theta (nu' F) from the point tables (chain_sens_th2_kernel) instead of chain_sens_th_kernel's reverse sweep
#endif
#ifndef MPCRL_CHAIN_MERGE_CALLS
#ifdef MPCRL_PROFILE_PHASES
#define MPCRL_CHAIN_MERGE_CALLS 0   // (the phase profile times the sweeps one by one)
#else
#define MPCRL_CHAIN_MERGE_CALLS 1   // predictor = factor + forward, corrector = backward + forward as ONE phase call each (half the callee-saved register traffic)
#endif
#endif
#ifndef MPCRL_CHAIN_FUSE_GT
#define MPCRL_CHAIN_FUSE_GT 1   // [B A]_k' nu_{k+1} (stationarity residual) formed by the direction pass from the columns it holds; 0: a stage pass over [B A]
#endif
#ifndef MPCRL_CHAIN_V2_ROUNDSTART
#define MPCRL_CHAIN_V2_ROUNDSTART 0
#endif
#ifndef MPCRL_CHAIN_V2_SENS_MAXNX
#define MPCRL_CHAIN_V2_SENS_MAXNX 33   // ... in the adjoint solves of the sensitivities (exact Hessian from the workspace, P_k streamed)
#endif

struct LargeSpec {
    int N, np, cost_kind, rk_steps, max_iter;
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

template <int NX, int NW>
struct ChainCfgStride {   // stage strides of the streamed blocks (ChainCfg below): whole 16-byte x 64-lane pieces
    static constexpr int AST = 128 * (((NX * NX + 1) / 2 + 63) / 64), BST = 128 * ((NX * NW / 2 + 63) / 64);
    static constexpr int PST = 128 * (((NX * (NX + 1) / 2 + 1) / 2 + 63) / 64);   // P_k in HBM: packed lower triangle
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
    size_t BA, r, q, dx, du, nuq, Dx, Du, Dnu, rg, rb, rt, Dg, lamw, tw, aff, P, p, K, L, kff, Acl, hb, ccv, cvec, Hex, term, ynu, state, Ydx, Ydu, Ydnu,
        term2, ptab, gtab, qvtab, G2, P2, hb2, minv2, mvu2, total;
    __host__ __device__ explicit LargeLayout(int N) {
        size_t o = 0;
        auto take = [&](size_t n) { size_t s = o; o += (n + 1) & ~(size_t)1; return s; };   // every array 16-byte aligned
        BA = take((size_t)N * NX * NW), r = take((size_t)N * NX), q = take((size_t)(N + 1) * NW);
        dx = take((size_t)(N + 1) * NX), du = take((size_t)N * NU), nuq = take((size_t)(N + 1) * NX);
        Dx = take((size_t)(N + 1) * NX + 2), Du = take((size_t)N * NU + 2), Dnu = take((size_t)(N + 1) * NX + 2);
        rg = take((size_t)(N + 1) * NW), rb = take((size_t)N * NX), rt = take((size_t)(N + 1) * NW), Dg = take((size_t)(N + 1) * NW);
        lamw = take((size_t)2 * (N + 1) * NW), tw = take((size_t)2 * (N + 1) * NW), aff = take((size_t)2 * (N + 1) * NW);
        P = take((size_t)(N + 1) * ChainCfgStride<NX, NW>::PST), p = take((size_t)(N + 1) * NX + 2), K = take((size_t)N * NU * NX), L = take((size_t)N * NU * NU);
        kff = take((size_t)N * NU + 2);   // (+ a dump slot each: masked lanes of the round-4 sweeps store there)
        Acl = take((size_t)N * ChainCfgStride<NX, NW>::BST), hb = take((size_t)N * NX), ccv = take((size_t)N * NX), cvec = take((size_t)N * NX);
        Hex = take((size_t)(N + 1) * NW * NW), term = take((size_t)N * NTD), ynu = take((size_t)(N + 1) * NX);
        state = take(16);   // ST_* below: the SQP loop's per-instance state between launches
        Ydx = take((size_t)NU * (N + 1) * NX + 2), Ydu = take((size_t)NU * N * NU), Ydnu = take((size_t)NU * (N + 1) * NX + 2);   // adjoint solutions
        term2 = take((size_t)NU * N * NTD);
        // per stage: coefficients of the 8 evaluation points of the RK4 map (chain_point_kernel), and the link Hessians of the adjoint
        ptab = take((size_t)N * 8 * M::NL * M::TAB2), gtab = take((size_t)N * 8 * M::NL * 6);
        qvtab = take((size_t)N * 8 * M::NL * 6);      // per evaluation point and link: force adjoint q (3), velocity difference dv (3) — chain_sens_mix2
        // round-4 sweeps: closed-loop blocks G_k, cost-to-go P_k (Omega register layout), hb_k = P_{k+1} b_k, R_k^-1, mv_u of the corrector
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
#ifdef MPCRL_PROFILE_PHASES
__device__ unsigned long long g_phase_ticks[16];
#endif

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

// tile shapes of the two stage GEMMs: one tile per lane, at most 64 tiles
template <class M>
struct DirCfg;
template <class M>
struct ChainCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU;
    static constexpr int TI = NX <= 9 ? 2 : (NX <= 21 ? 3 : 4);   // T = P [B A]   : TI x TJ outputs per lane
    static constexpr int TJ = NX <= 9 ? 3 : (NX <= 21 ? 4 : 6);
    static constexpr int TS = NX <= 9 ? 2 : (NX <= 21 ? 3 : 4);   // M = [B A]' T  : TS x TS outputs per lane, lower-triangular tile grid
    static constexpr int NTR = (NX + TI - 1) / TI, NTC = (NW + TJ - 1) / TJ, NMT = (NW + TS - 1) / TS;
    static constexpr int NTT = NTR * NTC, NMM = NMT * (NMT + 1) / 2;
    // the stage GEMMs on the matrix cores: 16-wide tiles of the [u; x] index (NT16 per side), of the state index (NTX), k-steps of 4
    static constexpr int NT16 = (NW + 15) / 16, NTX = (NX + 15) / 16, KS = (NX + 3) / 4;
    static constexpr bool V2_ = MPCRL_CHAIN_V2 != 0 && NX <= MPCRL_CHAIN_V2_MAXNX && NX <= MPCRL_CHAIN_V2_SENS_MAXNX;
    // (tilings of the round-3 sweeps: only the odd chain sizes are laid out for them; the even ones exist on the round-4 sweeps only)
    static_assert(V2_ || (NTT <= 64 && NMM <= 64), "one GEMM tile per lane");
    static_assert(V2_ || (NW % TJ == 0 && NW % TS == 0), "column tiles are full");
    static constexpr int NBA2 = (NX * NW / 2 + 63) / 64;    // 16-byte pieces per lane of one [B A] block
    static constexpr int NAC2 = ((NX * NX + 1) / 2 + 63) / 64;   // ... of one nx x nx block
    // stage strides of the streamed blocks in the workspace: whole pieces, so that the block copies need no tail predicate
    static constexpr int AST = 128 * NAC2;                  // P_k as a full block (its LDS staging area)
    // P_k is symmetric and travels through HBM as its packed lower triangle, (i, j) at i (i + 1) / 2 + j: half the bytes of the two
    // streams that carry it (factor sweep out, multiplier step of the forward sweep in)
    static constexpr int NPK = NX * (NX + 1) / 2, NPK2 = ((NPK + 1) / 2 + 63) / 64, PST = 128 * NPK2;
    static constexpr int BST = 128 * NBA2;                  // Acl_k (kept in the [B A] block shape)
#ifndef MPCRL_CHAIN_DEPTH
#define MPCRL_CHAIN_DEPTH (NX <= 21 ? 3 : 2)   // measured with the packed P stream: 3 gains 1 % at n_mass 5 and loses 7 % at n_mass 7 (registers)
#endif
#ifndef MPCRL_CHAIN_FDEPTH
#define MPCRL_CHAIN_FDEPTH 1
#endif
    static constexpr int DEPTH = MPCRL_CHAIN_DEPTH;         // stages in flight in the vector sweeps (<= 63 outstanding memory operations)
    static constexpr int FDEPTH = MPCRL_CHAIN_FDEPTH;       // ... in the factor sweep (a stage is ~2-6 k cycles of work there)
    static constexpr int UNR = NX <= 21 ? 3 : 1;            // inner indices per operand group of the stage GEMMs (two groups in registers)
    // LDS (doubles): the small vectors first (some of them live for the whole kernel), then the big region
    static constexpr int ev(int n) { return n + (n & 1); }
    static constexpr int oG = 0, oBB = oG + NW, oDG = oBB + ev(NX), oPV = oDG + NW, oCC = oPV + ev(NX), oMV = oCC + ev(NX), oK = oMV + NW,
                         oCK = oK + ev(NU * NX + NU), oSV = oCK + 64, oLB = oSV + 2 * ev(NX), oBig = oLB + 4 * NW + 8;
    // factor sweep: P_{k+1} and M share one region (P is dead once T = P [B A] and P b are formed, M once P_k is), [B A], T
    static constexpr int oP = oBig, oM = oBig, oBA = oBig + NW * NW, oT = oBA + NX * NW;
    // vector sweeps: Acl_k where P / M sit, P_k where [B A] sits; residual passes: Q where P / M sit, [B A] in place;
    // start of an SQP round: Q, then X - x_ss and U of the whole horizon (up to 64 stages)
    static constexpr int ASP = ev(NX * (NW + 1)) + 2;         // [B A]-shaped block published with the odd row stride NW + 1 (+ a dump slot)
    static constexpr int oA = oBig, oPk = oBig + (BST > ASP ? BST : ASP), oQ = oBig, oX = oBig + NW * NW, oU = oX + 64 * NX;
    static constexpr int BIG_F = NW * NW + 2 * NX * NW, BIG_R = NW * NW + 64 * NW;
    // round-4 sweeps: the Hessian table of the factor sweep, or two vectors of the whole horizon in Omega order (HBS doubles per
    // stage) — which depends on the horizon, so the kernels that run the solver take their LDS as a launch argument (lds_doubles)
    static constexpr bool V2 = MPCRL_CHAIN_V2 != 0 && NX <= MPCRL_CHAIN_V2_MAXNX;
    // [B A]_k' nu_{k+1} out of the direction pass (MPCRL_CHAIN_FUSE_GT).  Up to n_mass 5: at n_mass 7 the lane's four tangent arrays
    // already overflow the vector registers and the extra dot product costs the direction pass more (+450 us per solve) than the
    // stage pass it replaces (-380 us).
    static constexpr bool FUSE_GT = MPCRL_CHAIN_FUSE_GT != 0 && NX <= 21;
    static constexpr int BIG_O = BIG_F > BIG_R ? BIG_F : BIG_R;
    static constexpr int LDS_TOTAL = oBig + BIG_O;      // without the round-4 staging (DirCfg: table budget of the direction pass)
    __host__ __device__ static constexpr int lds_doubles(int N) {
        int big = BIG_O;
        if (V2) {
            const int tab = OmCfg<M>::RG * OmCfg<M>::NT * 64, vec = 2 * (N + 1) * OmCfg<M>::HBS, cst = (N + 1) * OmCfg<M>::HBS + tab;
            big = big > tab ? big : tab, big = big > vec ? big : vec, big = big > cst ? big : cst;
        }
        // the direction pass: its tables, the compact parameter copy and the multipliers of the whole horizon
        // (and the point pass's RK4 accumulators: NX per stage lane)
        const int dir = DirCfg<M>::CO + M::NTD + (FUSE_GT ? (N + 1) * NX : 0) + N * NX;
        big = big > dir ? big : dir;
        return oBig + big + (big & 1);
    }
    static_assert(LDS_TOTAL * 8 <= 40 * 1024, "four wavefronts per CU");
};

// Hessian source of the Riccati factorisation: the SQP uses c_k * (Q, R) from the parameter vector — constant per lane, kept in
// registers; the sensitivities use the exact Lagrangian Hessian blocks the kernel wrote to the workspace (read one stage ahead).
// Both hand out the lower-triangular 16 x 16 tiles of the [u; x] Hessian in the RESULT layout of v_mfma_f64_16x16x4 (register r of
// tile (tm, tj) at lane l = entry (16 tm + 4 r + l / 16, 16 tj + l % 16)), which is what the M = H + D + [B A]' T accumulators start from.
template <class M>
struct HessConst {
    using Cfg = ChainCfg<M>;
    static constexpr int NW = Cfg::NW, NT16 = Cfg::NT16, NLT = NT16 * (NT16 + 1) / 2;
    double h[NLT][4];     // unscaled, this lane's entries of the lower tiles
    const double *th;
    const double *sck;
    MPCRL_DI void begin(int lane) {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
            for (int tj = 0; tj <= tm; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * tm + 4 * r + lr, j = 16 * tj + lc;
                    h[tm * (tm + 1) / 2 + tj][r] = (i < NW && j < NW) ? M::hess(false, i < NW ? i : 0, j < NW ? j : 0, th) : 0.0;
                }
    }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned) { th = S.th, sck = S.sCK(); }   // (factor() calls begin())
    MPCRL_DI void prefetch(int) {}
    MPCRL_DI void advance(int) {}
    MPCRL_DI unsigned hex_offset() const { return 0u; }
    MPCRL_DI double tile(int k, int t, int r) const { return sck[k] * h[t][r]; }
    MPCRL_DI double term(int N, int i, int j) const { return sck[N] * M::Qs(th, i, j); }
};
template <class M>
struct HessGlobal {
    using Cfg = ChainCfg<M>;
    static constexpr int NW = Cfg::NW, NU = Cfg::NU, NT16 = Cfg::NT16, NLT = NT16 * (NT16 + 1) / 2;
    WsArr Hex;                      // [(N+1), NW, NW]
    int lr, lc;
    double hn[NLT][4], hc[NLT][4];
    MPCRL_DI void begin(int lane) { lr = lane >> 4, lc = lane & 15; }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned hex_off) { Hex = S.arr(hex_off); }
    MPCRL_DI void prefetch(int k) {
#pragma unroll
        for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
            for (int tj = 0; tj <= tm; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * tm + 4 * r + lr, j = 16 * tj + lc;
                    const bool in = i < NW && j < NW;
                    const double v = Hex[k * NW * NW + (in ? i * NW + j : 0)];      // rows of 16 lanes: 128-byte segments
                    hn[tm * (tm + 1) / 2 + tj][r] = in ? v : 0.0;
                }
    }
    // the tiles of stage k become current; those of stage k - 1 are requested (one stage ahead, independent of the operand ring)
    MPCRL_DI void advance(int k) {
#pragma unroll
        for (int t = 0; t < NLT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) hc[t][r] = hn[t][r];
        if (k > 0) prefetch(k - 1);
    }
    MPCRL_DI unsigned hex_offset() const { return Hex.off; }
    MPCRL_DI double tile(int, int t, int r) const { return hc[t][r]; }
    MPCRL_DI double term(int N, int i, int j) const { return Hex[N * NW * NW + (NU + i) * NW + NU + j]; }
};

// The same two Hessian sources in Omega coordinates (OmCfg): register (rg, tj) of a lane holds entry (slot 4 rg + lane / 16,
// slot 16 tj + lane % 16); slots past NW (and the vector column) are zero.
template <class M>
struct HessConst2 {
    using O = OmCfg<M>;
    const double *th;
    const double *sck;
    double *tab;        // LDS, [RG * NT][64]: this lane's entries, unscaled (12 - 27 registers a lane would otherwise hold per sweep)
    int lane_;
    MPCRL_DI void begin(int lane) {
        const int lr = lane >> 4, lc = lane & 15;
        lane_ = lane;
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < O::NW && c < O::NW;
                tab[(rg * O::NT + tj) * 64 + lane] = in ? M::hess(false, O::nat(in ? e : 0), O::nat(in ? c : 0), th) : 0.0;
            }
    }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned) { th = S.th, sck = S.sCK(), tab = S.lds + ChainCfg<M>::oBig; }
    MPCRL_DI void prefetch(int) {}
    MPCRL_DI void advance(int) {}
    MPCRL_DI double tile(int k, int rg, int tj) const { return sck[k] * tab[(rg * O::NT + tj) * 64 + lane_]; }
    MPCRL_DI double term(int N, int i, int j) const { return sck[N] * M::Qs(th, i, j); }
};
template <class M>
struct HessGlobal2 {
    using O = OmCfg<M>;
    static constexpr int NW = O::NW, NU = O::NU;
    WsArr Hex;                      // [(N+1), NW, NW], stage-vector order [u; x]
    int lr, lc;
    double hn[O::RG][O::NT], hc[O::RG][O::NT];
    MPCRL_DI void begin(int lane) { lr = lane >> 4, lc = lane & 15; }
    template <class S_>
    MPCRL_DI void init(const S_ &S, unsigned hex_off) { Hex = S.arr(hex_off); }
    MPCRL_DI void prefetch(int k) {
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < NW && c < NW;
                const double v = Hex[k * NW * NW + (in ? O::nat(e) * NW + O::nat(c) : 0)];
                hn[rg][tj] = in ? v : 0.0;
            }
    }
    MPCRL_DI void advance(int k) {
#pragma unroll
        for (int rg = 0; rg < O::RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < O::NT; ++tj) hc[rg][tj] = hn[rg][tj];
        if (k > 0) prefetch(k - 1);
    }
    MPCRL_DI double tile(int, int rg, int tj) const { return hc[rg][tj]; }
    MPCRL_DI double term(int N, int i, int j) const { return Hex[N * NW * NW + (NU + i) * NW + NU + j]; }
};
template <class HS>
struct HessV2;
template <class M>
struct HessV2<HessConst<M>> {
    using type = HessConst2<M>;
};
template <class M>
struct HessV2<HessGlobal<M>> {
    using type = HessGlobal2<M>;
};

template <class M>
struct ChainSolver {
    using Cfg = ChainCfg<M>;
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = 64;
    static constexpr int TI = Cfg::TI, TJ = Cfg::TJ, TS = Cfg::TS;
    static constexpr bool USE_V2 = MPCRL_CHAIN_V2 != 0 && NX <= MPCRL_CHAIN_V2_MAXNX, USE_V2_SENS = USE_V2 && NX <= MPCRL_CHAIN_V2_SENS_MAXNX;
    const LargeSpec *spp;   // kernel-argument copy of the problem (set-up only)
    const double *xs;       // x_ss (device)
    int N, lane;
    const double *th;   // full parameter vector of this instance
    bool qmode;
    // global (per instance)
    double *X, *U;
    WsArr NUv;             // NUv[k]: multiplier arriving at stage k, [(N+1)*NX] (index 0 unused)
    WsArr BA, r, q, dx, du, nuq, Dx, Du, Dnu, rg, rb, rt, Dg, lam, t, aff, P, p, K, L, kff, Acl, hb, ccv, cvec, state;
    WsArr G2, P2, hb2, minv2, mvu2;
    double *lds;
    // GEMM tiles of this lane
    int t_i0, t_j0, m_i0, m_j0;
    bool t_live, m_live, m_diag;
    // bounded rows: n0 / nm / ne coordinates carry a bound at stage 0 / 1..N-1 / N; coordinate lists in LDS (sidx)
    int n0, nm, ne, nrows;
    int *sidx;

    MPCRL_DI ChainSolver(const LargeSpec &s, int lane_) : spp(&s), xs(s.consts), N(s.N), lane(lane_) {}
    MPCRL_DI ChainSolver(const double *xs_, int N_, int lane_) : spp(nullptr), xs(xs_), N(N_), lane(lane_) {}   // inside a phase call: no set-up

#ifdef MPCRL_PROFILE_PHASES
    // per-wavefront tick counters in LDS (no global traffic inside the timed regions), flushed once by ph_flush()
    unsigned long long ph_t = 0;
    unsigned long long *ph_lds = nullptr;
    MPCRL_DI void ph0() { ph_t = clock64(); }
    MPCRL_DI void ph(int i) {
        const unsigned long long n_ = clock64();
        if (lane == 0) ph_lds[i] += n_ - ph_t;
        ph_t = n_;
    }
    MPCRL_DI void ph_init(unsigned long long *b) {
        ph_lds = b;
        if (lane < 16) b[lane] = 0;
        wave_sync();
    }
    MPCRL_DI void ph_flush() {
        wave_sync();
        if (lane < 16) atomicAdd(&g_phase_ticks[lane], ph_lds[lane]);
    }
#else
    MPCRL_DI void ph0() {}
    MPCRL_DI void ph(int) {}
    MPCRL_DI void ph_init(unsigned long long *) {}
    MPCRL_DI void ph_flush() {}
#endif

    MPCRL_DI double *sP() const { return lds + Cfg::oP; }
    MPCRL_DI double *sBA() const { return lds + Cfg::oBA; }
    MPCRL_DI double *sT() const { return lds + Cfg::oT; }
    MPCRL_DI double *sM() const { return lds + Cfg::oM; }
    MPCRL_DI double *sG() const { return lds + Cfg::oG; }
    MPCRL_DI double *sBB() const { return lds + Cfg::oBB; }
    MPCRL_DI double *sDG() const { return lds + Cfg::oDG; }
    MPCRL_DI double *sPV() const { return lds + Cfg::oPV; }
    MPCRL_DI double *sCC() const { return lds + Cfg::oCC; }
    MPCRL_DI double *sMV() const { return lds + Cfg::oMV; }
    MPCRL_DI double *sK() const { return lds + Cfg::oK; }
    MPCRL_DI double *sCK() const { return lds + Cfg::oCK; }
    MPCRL_DI double *sSV(int b) const { return lds + Cfg::oSV + b * Cfg::ev(NX); }
    MPCRL_DI double *sLB() const { return lds + Cfg::oLB; }          // lb, ub (stages 1..N-1), lbe, ube: NW each; lb0, ub0: 4 each
    MPCRL_DI double *sA() const { return lds + Cfg::oA; }
    MPCRL_DI double *sPk() const { return lds + Cfg::oPk; }

    MPCRL_DI double ck(int k) const { return sCK()[k]; }
    MPCRL_DI double ck_eval(int k) const {
        const LargeSpec &sp = *spp;
        if (sp.cost_kind == 0) return k == N ? 1.0 : sp.dT;                                            // nlp.py:1044-1055
        return k == 0 ? sp.dT : (k == N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);   // nlp.py:1083-1091
    }
    // bounds out of LDS (a lane-dependent index into kernel arguments would be copied to scratch)
    MPCRL_DI double lbv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sLB()[4 * NW + i] : -1e30;
        if (k == N) return i >= NU ? sLB()[2 * NW + i] : -1e30;
        return sLB()[i];
    }
    MPCRL_DI double ubv(int k, int i) const {
        if (k == 0) return (i < NU && !qmode) ? sLB()[4 * NW + 4 + i] : 1e30;
        if (k == N) return i >= NU ? sLB()[3 * NW + i] : 1e30;
        return sLB()[NW + i];
    }
    MPCRL_DI bool has(int sd, int k, int i) const { return sd ? ubv(k, i) < NO_BOUND : lbv(k, i) > -NO_BOUND; }
    MPCRL_DI bool fixedc(int k, int i) const { return k == 0 && (i >= NU || qmode); }
    MPCRL_DI bool skipc(int k, int i) const { return k == N && i < NU; }
    MPCRL_DI double vc(int k, int i) const { return i < NU ? (k < N ? U[k * NU + i] : 0.0) : X[k * NX + i - NU]; }
    MPCRL_DI double dvc(const WsArr &ax, const WsArr &au, int k, int i) const {
        return i < NU ? (k < N ? au[k * NU + i] : 0.0) : ax[k * NX + i - NU];
    }
    MPCRL_DI double bslack(int sd, int k, int i, double v) const { return sd ? ubv(k, i) - v : v - lbv(k, i); }
    // row r of the compact list of bounded coordinates -> (stage, coordinate)
    MPCRL_DI void row_of(int r_, int &k, int &i) const {
        if (r_ < n0) {
            k = 0, i = sidx[r_];
        } else if (r_ < n0 + (N - 1) * nm) {
            const int q_ = r_ - n0;
            const int kk = q_ / nm;
            k = kk + 1, i = sidx[64 + q_ - kk * nm];
        } else
            k = N, i = sidx[128 + r_ - n0 - (N - 1) * nm];
    }
    MPCRL_DI double &LAM(int sd, int e) { return lam[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &TT(int sd, int e) { return t[sd * (N + 1) * NW + e]; }
    MPCRL_DI double &AFF(int sd, int e) { return aff[sd * (N + 1) * NW + e]; }

    // 16-byte coalesced moves of one stage block: workspace -> registers -> LDS (n2 = number of 16-byte pieces)
    template <int NV>
    MPCRL_DI void blk_load(const WsArr src, int n2, d2_t (&rr)[NV], int lane) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int e2 = lane + 64 * s;
            (void)n2;
            rr[s] = *(const d2_t *)&src[2 * e2];   // past the block: the next array of the workspace (never used)
        }
    }
    // full-block offsets (i NX + j) of the packed entries 2 (lane + 64 s) + h this lane carries; entries past the triangle -> dump words
    // behind the block.  tr = true: the mirrored offsets (j NX + i).
    MPCRL_DI void packed_offsets(int (&off)[Cfg::NPK2][2], bool tr) const {
#pragma unroll
        for (int s_ = 0; s_ < Cfg::NPK2; ++s_)
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
                const int q = 2 * (lane + 64 * s_) + h_;
                int i = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
                if ((i + 1) * (i + 2) / 2 <= q) ++i;
                if (i * (i + 1) / 2 > q) --i;
                const int j = q - i * (i + 1) / 2;
                off[s_][h_] = q < Cfg::NPK ? (tr ? j * NX + i : i * NX + j) : NX * NX + h_;
            }
    }
    template <int NV>
    MPCRL_DI void blk_to_lds(double *dst, int n2, const d2_t (&rr)[NV], int lane) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int e2 = lane + 64 * s;
            (void)n2;
            *(d2_t *)(dst + 2 * e2) = rr[s];       // the LDS regions leave room for 128 * NV doubles
        }
    }

    // ---- per-lane indices of the solver (a pure function of the lane and of the row counts the set-up left in LDS): what a phase
    // call re-derives on entry instead of receiving it
    MPCRL_DI void setup_lane(double *lds_, int *sidx_, bool read_counts) {
        lds = lds_, sidx = sidx_;
        {   // T tile: row-major tile index
            const int tr = lane / Cfg::NTC, tc = lane - tr * Cfg::NTC;
            t_live = lane < Cfg::NTT;
            t_i0 = t_live ? tr * TI : 0, t_j0 = t_live ? tc * TJ : 0;
        }
        {   // M tile: lower-triangular tile grid, tile index l = ti (ti + 1) / 2 + tj
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= lane) ++ti;
            const int tj = lane - ti * (ti + 1) / 2;
            m_live = lane < Cfg::NMM;
            m_i0 = m_live ? ti * TS : 0, m_j0 = m_live ? tj * TS : 0;
            m_diag = ti == tj;
        }
        if (read_counts) {
            n0 = (int)rfl((unsigned)sidx[192]), nm = (int)rfl((unsigned)sidx[193]), ne = (int)rfl((unsigned)sidx[194]);
            nrows = n0 + (N - 1) * nm + ne;
        }
    }
    // ---- one-off set-up: constants into LDS, GEMM tile of this lane, list of bounded coordinates
    MPCRL_DI void setup(double *lds_, int *sidx_) {
        const LargeSpec &sp = *spp;
        setup_lane(lds_, sidx_, false);
        if (lane <= N) sCK()[lane] = ck_eval(lane);
        if (lane == 0) {   // one lane, compile-time indices: the kernel arguments stay scalar operands
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                sLB()[i] = sp.lb[i], sLB()[NW + i] = sp.ub[i];
                sLB()[2 * NW + i] = i >= NU ? sp.lbe[i >= NU ? i - NU : 0] : -1e30, sLB()[3 * NW + i] = i >= NU ? sp.ube[i >= NU ? i - NU : 0] : 1e30;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sLB()[4 * NW + i] = sp.lb0[i], sLB()[4 * NW + 4 + i] = sp.ub0[i];
        }
        wave_sync();
        if (lane == 0) {
            int a = 0, b = 0, c = 0;
            for (int i = 0; i < NW; ++i) {
                if (i < NU && !qmode && (sLB()[4 * NW + (i < 4 ? i : 0)] > -NO_BOUND || sLB()[4 * NW + 4 + (i < 4 ? i : 0)] < NO_BOUND)) sidx[a++] = i;
                if (sLB()[i] > -NO_BOUND || sLB()[NW + i] < NO_BOUND) sidx[64 + b++] = i;
                if (i >= NU && (sLB()[2 * NW + i] > -NO_BOUND || sLB()[3 * NW + i] < NO_BOUND)) sidx[128 + c++] = i;
            }
            sidx[192] = a, sidx[193] = b, sidx[194] = c;
        }
        wave_sync();
        n0 = sidx[192], nm = sidx[193], ne = sidx[194];
        nrows = n0 + (N - 1) * nm + ne;
    }

    // ([B A]_k' nu_{k+1} - [0; nu_k])_i
    MPCRL_DI double GTnu(const WsArr &nu, int k, int i) const {
        double a = 0.0;
        if (k < N) {
            const WsArr Bk = BA + (k * NX * NW + i);
            for (int m = 0; m < NX; ++m) a = fma(Bk[m * NW], nu[(k + 1) * NX + m], a);
        }
        if (i >= NU && k > 0) a -= nu[k * NX + i - NU];
        return a;
    }

    // ---- start of an SQP round: q = c_k grad l_k, the cost, and the four NLP residual norms (stationarity, equality,
    // inequality, complementarity).  Q (symmetrised), X - x_ss and U of the whole horizon are staged in LDS; the stationarity
    // residual q + [B A]' nu_{k+1} - [0; nu_k] -+ lam is a stage-serial pass with [B A]_k staged through LDS (coalesced).
    MPCRL_DI double round_start(const double *x0, const double *u0f, double *res) {
        const int ne = (N + 1) * NW;
        double *lQ = lds + Cfg::oQ, *lX = lds + Cfg::oX, *lU = lds + Cfg::oU;
        for (int e = lane; e < NX * NX; e += NT) lQ[e] = M::Qs(th, e / NX, e % NX);
        batched_pass<8>((N + 1) * NX, lane, [&](int e) { return X[e]; }, [&](int e, double v) { lX[e] = v - xs[e % NX]; });
        batched_pass<2>(N * NU, lane, [&](int e) { return U[e]; }, [&](int e, double v) { lU[e] = v; });
        wave_sync();
        double val = 0.0;
        batched_pass<4>(ne, lane, [&](int e) { return Pair2{lam[e], lam[ne + e]}; }, [&](int e, const Pair2 &lm) {
            const int k = e / NW, i = e - k * NW;
            const bool term = k == N;
            double a = 0.0, v = 0.0;
            if (i < NU) {
                if (!term) {
#pragma unroll
                    for (int j = 0; j < NU; ++j) a = fma(M::Rs(th, i, j), lU[k * NU + j], a);
                    v = lU[k * NU + i];
                }
            } else {
                const double *qr = lQ + (i - NU) * NX, *xk = lX + k * NX;
                a = lds_dot<NX>(qr, 1, xk, 0.0);
                v = xk[i - NU];
            }
            const double qe = ck(k) * a;
            q[e] = qe;
            val = fma(0.5 * qe, v, val);
            double g = qe;
            if (!skipc(k, i)) {
                if (has(0, k, i)) g -= lm.a;
                if (has(1, k, i)) g += lm.b;
            }
            rg[e] = g;   // q -+ lam: the stage pass below adds the multiplier terms of the dynamics
        });
        wave_sync();
        double rs = 0, re = 0, ri = 0, rc = 0;
        if constexpr (USE_V2 && MPCRL_CHAIN_V2_ROUNDSTART) {   // (measured slower than the staged pass below: 77 vs 64 us per call at n_mass 5)
            // [B A]_k' nu_{k+1} of all stages by the MFMA pass (the staged Q, X, U above are dead), then one pass over the entries
            double *const lnu = lds + Cfg::oBig, *const ly = lnu + (N + 1) * OmCfg<M>::HBS;
            wt_nu_pass(NUv, lnu, ly);
            batched_pass<4>(ne, lane,
                            [&](int e) {
                                const int k = e / NW, i = e - k * NW;
                                return Pair2{rg[e], (i >= NU && k > 0) ? NUv[k * NX + i - NU] : 0.0};
                            },
                            [&](int e, const Pair2 &v) {
                                const int k = e / NW, i = e - k * NW;
                                const double a = v.a - v.b + (k < N ? ly[k * OmCfg<M>::HBS + om_slot(i)] : 0.0);
                                if (!fixedc(k, i) && !skipc(k, i)) rs = fmax(rs, fabs(a));
                            });
        } else if constexpr (Cfg::FUSE_GT) {
            // [B A]_k' nu_{k+1} arrives in rt from the direction pass of this round (chain_dir_pass): one pass over the entries
            batched_pass<4>(ne, lane,
                            [&](int e) {
                                const int k = e / NW, i = e - k * NW;
                                return Quad4{rg[e], (i >= NU && k > 0) ? NUv[k * NX + i - NU] : 0.0, k < N ? rt[e] : 0.0, 0.0};
                            },
                            [&](int e, const Quad4 &v) {
                                const int k = e / NW, i = e - k * NW;
                                if (k == N && i < NU) return;           // (no controls at the terminal stage)
                                const double a = (v.a - v.b) + v.c;
                                const bool fx = k < N && fixedc(k, i);
                                if (!fx) rs = fmax(rs, fabs(a));
                                // the vector itself stays in rg: it IS the stationarity residual the next QP starts from (qp_start_residuals)
                                if (USE_V2) rg[e] = fx ? 0.0 : a;
                            });
        } else {
            d2_t nB[Cfg::DEPTH][Cfg::NBA2];
            double ng[Cfg::DEPTH], nn[Cfg::DEPTH], no[Cfg::DEPTH];
            double *lBA = sBA(), *lnu = sBB();
            const int lj = lane < NW ? lane : 0, lx = lane < NX ? lane : 0;
            staged_loop<Cfg::DEPTH>(
                N,
                [&](int k, auto sl) {
                    constexpr int d = decltype(sl)::value;
                    blk_load(BA + k * NX * NW, NX * NW / 2, nB[d], lane);
                    ng[d] = rg[k * NW + lj], nn[d] = NUv[(k + 1) * NX + lx];
                    {
                        const bool c_ = lane >= NU && lane < NW && k > 0;
                        const double t_ = NUv[c_ ? k * NX + lj - NU : 0];
                        no[d] = c_ ? t_ : 0.0;
                    }
                },
                [&](int k, auto sl, auto refill) {
                    constexpr int d = decltype(sl)::value;
                    blk_to_lds(lBA, NX * NW / 2, nB[d], lane);
                    if (lane < NX) lnu[lane] = nn[d];
                    double a = ng[d] - no[d];
                    refill();
                    wave_sync();
                    a = lds_dot<NX>(lBA + lj, NW, lnu, a);
                    if (lane < NW && !fixedc(k, lane)) rs = fmax(rs, fabs(a));
                    // the vector itself stays in rg: it IS the stationarity residual the next QP starts from (qp_start_residuals)
                    if (USE_V2 && lane < NW) rg[k * NW + lane] = fixedc(k, lane) ? 0.0 : a;
                    wave_sync();
                });
            if (lane >= NU && lane < NW) {
                const double a = rg[N * NW + lane] - NUv[N * NX + lane - NU];
                rs = fmax(rs, fabs(a));
                if (USE_V2) rg[N * NW + lane] = a;
            }
        }
        for (int r_ = lane; r_ < nrows; r_ += NT) {
            int k, i;
            row_of(r_, k, i);
            const int e = k * NW + i;
            const double v = vc(k, i);
            if (has(0, k, i)) {
                const double h = lbv(k, i) - v;
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[e] * h));
            }
            if (has(1, k, i)) {
                const double h = v - ubv(k, i);
                ri = fmax(ri, h), rc = fmax(rc, fabs(lam[ne + e] * h));
            }
        }
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int, double v) { re = fmax(re, fabs(v)); });
        if (lane < NX) re = fmax(re, fabs(X[lane] - x0[lane]));
        if (qmode && lane < NU) re = fmax(re, fabs(U[lane] - u0f[lane]));
        res[0] = wave_max(rs), res[1] = wave_max(re), res[2] = wave_max(ri), res[3] = wave_max(rc);
        return wave_sum(val);
    }

    // ---- residuals of the QP at (dx, du, nuq, lam): rb = r + [B A] dv - dx+  and  rg = q + H dv + [B A]' nuq+ - [0; nuq] -+ lam.
    // Same stage pass: [B A]_k through LDS, lane i < NX takes row i (read skewed by i so that the row stride NW does not
    // collide on LDS banks), lane j < NW takes column j and its row of the stage Hessian.  Returns the inf-norm.
    MPCRL_DI double qp_residuals() {
        const int ne = (N + 1) * NW;
        double *lQ = lds + Cfg::oQ, *lBA = sBA(), *ldv = sG(), *lnu = sBB();
        for (int e = lane; e < NX * NX; e += NT) lQ[e] = M::Qs(th, e / NX, e % NX);
        batched_pass<4>(ne, lane, [&](int e) { return Quad4{q[e], lam[e], lam[ne + e], 0.0}; },
                        [&](int e, const Quad4 &v) {
                            const int k = e / NW, i = e - k * NW;
                            double g = v.a;
                            if (!skipc(k, i)) {
                                if (has(0, k, i)) g -= v.b;
                                if (has(1, k, i)) g += v.c;
                            }
                            rg[e] = g;
                        });
        double Rrow[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) Rrow[j] = lane < NU ? M::Rs(th, lane < NU ? lane : 0, j) : 0.0;
        wave_sync();
        double rloc = 0.0;
        d2_t nB[Cfg::DEPTH][Cfg::NBA2];
        double nv[Cfg::DEPTH], nn[Cfg::DEPTH], nxn[Cfg::DEPTH], nr[Cfg::DEPTH], ng[Cfg::DEPTH], no[Cfg::DEPTH];
        const int lj = lane < NW ? lane : 0, lx = lane < NX ? lane : 0;
        const double *qr = lQ + (lane >= NU && lane < NW ? lane - NU : 0) * NX;
        // [B A]_k is published with an odd row stride (NW + 1): lane i reads row i, and NW doubles between rows would put a
        // whole lane group on four LDS banks
        constexpr int NWP = NW + 1;
        int pdst[Cfg::NBA2][2];
#pragma unroll
        for (int s_ = 0; s_ < Cfg::NBA2; ++s_)
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
                const int e = 2 * (lane + 64 * s_) + h_;
                pdst[s_][h_] = e < NX * NW ? (e / NW) * NWP + e % NW : NX * NWP + (e & 1);   // past the block: a dump slot
            }
        staged_loop<Cfg::DEPTH>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
                blk_load(BA + k * NX * NW, NX * NW / 2, nB[d], lane);
                nv[d] = WsArr{dx.base, lane < NU ? du.off + k * NU + lj : dx.off + k * NX + (lane < NW ? lane - NU : 0)}[0];
                nn[d] = nuq[(k + 1) * NX + lx], nxn[d] = dx[(k + 1) * NX + lx], nr[d] = r[k * NX + lx];
                ng[d] = rg[k * NW + lj];
                {
                    const bool c_ = lane >= NU && lane < NW && k > 0;
                    const double t_ = nuq[c_ ? k * NX + lj - NU : 0];
                    no[d] = c_ ? t_ : 0.0;
                }
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
#pragma unroll
                for (int s_ = 0; s_ < Cfg::NBA2; ++s_) {
                    lBA[pdst[s_][0]] = nB[d][s_].x;
                    lBA[pdst[s_][1]] = nB[d][s_].y;
                }
                if (lane < NW) ldv[lane] = nv[d];
                if (lane < NX) lnu[lane] = nn[d];
                double a = nr[d] - nxn[d], g = ng[d] - no[d];
                refill();
                wave_sync();
                a = lds_dot<NW>(lBA + lx * NWP, 1, ldv, a);          // row lx of [B A] times dv
                g = lds_dot<NX>(lBA + lj, NWP, lnu, g);              // column lj of [B A] times nuq+
                double hd = lds_dot<NX>(qr, 1, ldv + NU, 0.0);       // row of Q times dx (lanes < NU: overwritten below)
                if (lane < NU) {
                    hd = 0.0;
#pragma unroll
                    for (int l = 0; l < NU; ++l) hd = fma(Rrow[l], ldv[l], hd);
                }
                g = fma(ck(k), hd, g);
                if (fixedc(k, lane)) g = 0.0;
                if (lane < NX) rb[k * NX + lane] = a, rloc = fmax(rloc, fabs(a));
                if (lane < NW) rg[k * NW + lane] = g, rloc = fmax(rloc, fabs(g));
                wave_sync();
            });
        {   // terminal stage: no control, no dynamics leaving it
            if (lane < NX) ldv[NU + lane] = dx[N * NX + lane];
            wave_sync();
            double g = 0.0;
            if (lane >= NU && lane < NW) {
                const double hd = lds_dot<NX>(qr, 1, ldv + NU, 0.0);
                g = fma(ck(N), hd, rg[N * NW + lane] - nuq[N * NX + lane - NU]);
            }
            if (lane < NW) rg[N * NW + lane] = g, rloc = fmax(rloc, fabs(g));
        }
        wave_sync();
        return rloc;
    }


    // ---- Riccati factor sweep with the vector recursion of the first right-hand side riding along.
    // HS: Hessian source; g: modified gradient [(N+1)*NW]; bb: dynamics offsets [N*NX] or null (= 0).
    // Leaves in the workspace: P_k, p_k, K_k, L_k, kff_k, Acl_k = A_k - B_k K_k, hb_k = P_{k+1} b_k.
    template <class HS>
    MPCRL_DI bool factor(HS &hs, const WsArr g, const WsArr bb) {
        hs.begin(lane);
        bool ok = true;
        double *const lP = sP(), *const lBA = sBA(), *const lT = sT(), *const lM = sM();
        constexpr int AST = Cfg::AST, FD = Cfg::FDEPTH;
        // terminal stage
        for (int e = lane; e < NX * NX; e += NT) {
            const int i = e / NX, j = e - i * NX;
            const double v = hs.term(N, i > j ? i : j, i > j ? j : i) + (i == j ? Dg[N * NW + NU + i] : 0.0);
            lP[e] = v;
            if (j <= i) P[N * Cfg::PST + i * (i + 1) / 2 + j] = v;
        }
        int pko[Cfg::NPK2][2];
        packed_offsets(pko, false);
        if (lane < NX) {
            const double v = g[N * NW + NU + lane];
            sPV()[lane] = v;
            p[N * NX + lane] = v;
        }
        d2_t nB[FD][Cfg::NBA2];
        double ng[FD], nbb[FD], nDg[FD];
        const int lj = lane < NW ? lane : 0, lx = lane < NX ? lane : 0;
        hs.prefetch(N - 1);   // the Hessian source keeps one stage ahead itself (HessGlobal::advance)
        staged_loop<FD>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                blk_load(BA + k * NX * NW, NX * NW / 2, nB[d], lane);
                ng[d] = g[k * NW + lj], nDg[d] = Dg[k * NW + lj];
                nbb[d] = bb[k * NX + lx];
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const bool pin = k == 0 && qmode;
                // ---- publish the stage operands, start fetching a later stage's
                blk_to_lds(lBA, NX * NW / 2, nB[d], lane);
                if (lane < NW) sG()[lane] = ng[d], sDG()[lane] = nDg[d];
                if (lane < NX) sBB()[lane] = nbb[d];
                hs.advance(k);
                refill();
                wave_sync();
                // ---- T = P [B A] and M = H + D + [B A]' T on the matrix cores (v_mfma_f64_16x16x4; operand / result layouts measured,
                // profiles/microbench/mfma_f64_16x16x4_probe.hip: A(i, k) and B(k, j) both sit at lane 16 k + (i | j), result register r
                // of a tile holds (4 r + lane / 16, lane % 16)).  Per k-step of 4 rows one LDS read per tile brings
                //   pa = P[4 ks + lane / 16][16 ti + lane % 16]    A operand of T (P is symmetric: read along its rows, conflict-free)
                //   bb = [B A][4 ks + lane / 16][16 tj + lane % 16]  B operand of T — and, unchanged, the A operand [B A]' of M;
                // T comes out in the result layout, whose register r IS the B operand of k-step r of the next product, so T never
                // touches LDS.  (The register-tiled VALU version read TI + TJ operands out of LDS per 12 FMAs and ran at a fifth of the
                // FMA issue rate; fp64 MFMA has the same peak as the vector ALU, the gain is the operand traffic.)
                {
                    typedef double d4_t __attribute__((ext_vector_type(4)));
                    constexpr int NT16 = Cfg::NT16, NTX = Cfg::NTX, KS = Cfg::KS;
                    const int lr = lane >> 4, lc = lane & 15;
                    double pa[NTX][KS], bb_[NT16][KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int kr = 4 * ks + lr;
                        const bool krow = kr < NX;
#pragma unroll
                        for (int ti = 0; ti < NTX; ++ti) {
                            const int c = 16 * ti + lc;
                            const double v = lP[(krow ? kr : 0) * NX + (c < NX ? c : 0)];
                            pa[ti][ks] = (krow && c < NX) ? v : 0.0;
                        }
#pragma unroll
                        for (int tj = 0; tj < NT16; ++tj) {
                            const int c = 16 * tj + lc;
                            const double v = lBA[(krow ? kr : 0) * NW + (c < NW ? c : 0)];
                            bb_[tj][ks] = (krow && c < NW) ? v : 0.0;
                        }
                    }
                    const double a = lds_dot<NX>(lP + lx * NX, 1, sBB(), 0.0);      // (P b)_lane, before M overwrites P
                    MPCRL_SCHED_FENCE();
                    d4_t Tt[NTX][NT16];
#pragma unroll
                    for (int ti = 0; ti < NTX; ++ti)
#pragma unroll
                        for (int tj = 0; tj < NT16; ++tj) {
                            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ti][ks], bb_[tj][ks], acc, 0, 0, 0);
                            Tt[ti][tj] = acc;
                        }
                    if (lane < NX) {
                        sCC()[lane] = sPV()[lane] + a;
                        hb[k * NX + lane] = a;
                    }
                    ph(10);
                    d4_t Mt[NT16 * (NT16 + 1) / 2];
#pragma unroll
                    for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
                        for (int tj = 0; tj <= tm; ++tj) {
                            const int t_ = tm * (tm + 1) / 2 + tj;
                            d4_t acc;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * tm + 4 * r + lr;
                                acc[r] = hs.tile(k, t_, r) + ((tm == tj && i == 16 * tj + lc && i < NW) ? sDG()[i < NW ? i : 0] : 0.0);
                            }
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bb_[tm][ks], Tt[ks / 4][tj][ks % 4], acc, 0, 0, 0);
                            Mt[t_] = acc;
                        }
                    wave_sync();                                                      // cc is complete
                    const double mv_l = lds_dot<NX>(lBA + lj, NW, sCC(), sG()[lj]);   // mv = g + [B A]' cc
                    // M overwrites P_{k+1} (same LDS region): every lane is past its reads of P (operands and P b are in registers).
                    // Lower triangle, mirrored: exactly symmetric.
                    wave_sync();
#pragma unroll
                    for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
                        for (int tj = 0; tj <= tm; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * tm + 4 * r + lr, j = 16 * tj + lc;
                                if (i < NW && j <= i) {
                                    const double v = Mt[tm * (tm + 1) / 2 + tj][r];
                                    lM[i * NW + j] = v;
                                    lM[j * NW + i] = v;
                                }
                            }
                    if (lane < NW) sMV()[lane] = mv_l;
                }
                wave_sync();
                ph(11);
                // ---- Cholesky of the control block (every lane, redundantly): L lower with inverted diagonal
                double Lc[NU][NU];
                {
                    bool okc = true;
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double a = lM[i * NW + j];
#pragma unroll
                            for (int m = 0; m < j; ++m) a -= Lc[i][m] * Lc[j][m];
                            if (i == j) {
                                okc = okc && (a > 0.0);
                                Lc[i][i] = 1.0 / sqrt(a);
                            } else
                                Lc[i][j] = a * Lc[j][j];
                        }
                    ok = ok && (okc || pin);
                }
                // K columns (lanes j < NX) and the feed-forward (lane NX): solve L L' z = rhs
                if (lane <= NX) {
                    const int j = lane;
                    double y[NU], z[NU];
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        double a = j < NX ? lM[i * NW + NU + j] : sMV()[i];   // S(i, j) read along row i of the (exactly symmetric) M: conflict-free
#pragma unroll
                        for (int m = 0; m < i; ++m) a -= Lc[i][m] * y[m];
                        y[i] = a * Lc[i][i];
                    }
#pragma unroll
                    for (int i = NU - 1; i >= 0; --i) {
                        double a = y[i];
#pragma unroll
                        for (int m = i + 1; m < NU; ++m) a -= Lc[m][i] * z[m];
                        z[i] = a * Lc[i][i];
                    }
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        const double v = pin ? 0.0 : z[i];
                        if (j < NX) {
                            sK()[i * NX + j] = v;
                            K[(k * NU + i) * NX + j] = v;
                        } else {
                            sK()[NU * NX + i] = v;
                            kff[k * NU + i] = v;
                        }
                    }
                }
                if (lane < NU * NU) {
                    const int i = lane / NU, j = lane - i * NU;
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < NU; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b)
                            if (a == i && b == j) v = Lc[a][b];
                    L[k * NU * NU + lane] = pin ? 0.0 : v;
                }
                wave_sync();
                ph(12);
                // ---- P_k = Q - S' K, p_k = mv_x - K' mv_u, Acl = A - B K: two rank-NU updates, on the matrix cores as well.  In stage-vector
                // coordinates (v = [u; x]) with K' = [0 | K]:  P' = M - M(0..NU, .)' K' on the lower tiles of M's own tiling (the x-x block
                // of P' is P_k), [B Acl] = [B A] - B K'.  One v_mfma_f64_16x16x4 per 16 x 16 tile (k = the NU rows of K, padded to 4); the
                // operands are single LDS reads in their register layout.  (The column-per-lane VALU loop it replaces up to n_mass 5 — kept below
                // for n_mass 7 — is 170 LDS reads and 63 writes per lane for 130 FMAs: LDS-instruction bound, 13 % of the kernel; 12.63 -> 12.24 ms.)
                // P_k is written lower + mirrored into T (exactly symmetric), Acl replaces A inside [B A]; both blocks then leave for HBM
                // as 16-byte coalesced copies.
                if constexpr (NW <= 24) {
                    typedef double d4_t __attribute__((ext_vector_type(4)));
                    constexpr int NT16 = Cfg::NT16, NTX = Cfg::NTX;
                    const int lr = lane >> 4, lc = lane & 15;
                    const double *lK = sK();
                    double kb[NT16], sa[NT16], bm[NTX];
#pragma unroll
                    for (int t = 0; t < NT16; ++t) {
                        const int c = 16 * t + lc;
                        const bool onk = lr < NU && c >= NU && c < NW, ons = lr < NU && c < NW;
                        const double vk = lK[(onk ? lr : 0) * NX + (onk ? c - NU : 0)], vs = lM[(ons ? lr : 0) * NW + (ons ? c : 0)];
                        kb[t] = onk ? -vk : 0.0;     // B operand of both products: -K'(k = lane / 16, column c)
                        sa[t] = ons ? vs : 0.0;      // A operand of P': M(k, row c) = S'(c, k)
                    }
#pragma unroll
                    for (int t = 0; t < NTX; ++t) {
                        const int i = 16 * t + lc;
                        const bool on = lr < NU && i < NX;
                        const double v = lBA[(on ? i : 0) * NW + (on ? lr : 0)];
                        bm[t] = on ? v : 0.0;        // A operand of the second product: B(row i, k)
                    }
                    // the C operands of both products are fetched before the first MFMA issues: their LDS latency is paid once
                    d4_t Pc[NT16 * (NT16 + 1) / 2], Ac[NTX][NT16];
#pragma unroll
                    for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
                        for (int tj = 0; tj <= tm; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int a_ = 16 * tm + 4 * r + lr, b_ = 16 * tj + lc;
                                const bool on = a_ < NW && b_ < NW;
                                const double v = lM[(on ? a_ : 0) * NW + (on ? b_ : 0)];
                                Pc[tm * (tm + 1) / 2 + tj][r] = on ? v : 0.0;
                            }
#pragma unroll
                    for (int ti = 0; ti < NTX; ++ti)
#pragma unroll
                        for (int tb = 0; tb < NT16; ++tb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * ti + 4 * r + lr, b_ = 16 * tb + lc;
                                const bool on = i < NX && b_ < NW;
                                const double v = lBA[(on ? i : 0) * NW + (on ? b_ : 0)];
                                Ac[ti][tb][r] = on ? v : 0.0;
                            }
                    MPCRL_SCHED_FENCE();
#pragma unroll
                    for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
                        for (int tj = 0; tj <= tm; ++tj) {
                            const int t_ = tm * (tm + 1) / 2 + tj;
                            Pc[t_] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[tm], kb[tj], Pc[t_], 0, 0, 0);
                        }
#pragma unroll
                    for (int ti = 0; ti < NTX; ++ti)
#pragma unroll
                        for (int tb = 0; tb < NT16; ++tb) Ac[ti][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bm[ti], kb[tb], Ac[ti][tb], 0, 0, 0);
                    wave_sync();   // every lane is past its operand reads of [B A] (one wavefront: LDS operations are performed in order)
#pragma unroll
                    for (int tm = 0; tm < NT16; ++tm)
#pragma unroll
                        for (int tj = 0; tj <= tm; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int a_ = 16 * tm + 4 * r + lr, b_ = 16 * tj + lc;
                                if (a_ < NW && b_ >= NU && b_ <= a_) {
                                    const double v = Pc[tm * (tm + 1) / 2 + tj][r];
                                    lT[(a_ - NU) * NX + (b_ - NU)] = v;
                                    lT[(b_ - NU) * NX + (a_ - NU)] = v;
                                }
                            }
#pragma unroll
                    for (int ti = 0; ti < NTX; ++ti)
#pragma unroll
                        for (int tb = 0; tb < NT16; ++tb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {   // the B columns stay as they are
                                const int i = 16 * ti + 4 * r + lr, b_ = 16 * tb + lc;
                                if (i < NX && b_ >= NU && b_ < NW) lBA[i * NW + b_] = Ac[ti][tb][r];
                            }
                } else {
                    // n_mass 7: the matrix-core version needs 9 + 6 result tiles next to a register file that is already full (measured: 39.1 ->
                    // 40.0 ms with the two products one after the other, 44.9 ms fused); lane j takes COLUMN j of both updates — the lanes of a
                    // group read consecutive LDS words, S(., i) and B(i, .) are wave-uniform reads, K(., j) sits in three registers
                    const double *lK = sK();
                    double Kj[NU];
#pragma unroll
                    for (int m = 0; m < NU; ++m) Kj[m] = lK[m * NX + lx];
                    constexpr int IC = NX % 7 == 0 ? 7 : 3;
                    static_assert(NX % IC == 0, "row chunks");
                    for (int i0 = 0; i0 < NX; i0 += IC) {
                        double mq[IC], aq[IC], si[IC][NU], bi[IC][NU];
#pragma unroll
                        for (int ii = 0; ii < IC; ++ii) {
                            const int i = i0 + ii;
                            mq[ii] = lM[(NU + i) * NW + NU + lx], aq[ii] = lBA[i * NW + NU + lx];
#pragma unroll
                            for (int m = 0; m < NU; ++m) si[ii][m] = lM[m * NW + NU + i], bi[ii][m] = lBA[i * NW + m];
                        }
                        MPCRL_SCHED_FENCE();
#pragma unroll
                        for (int ii = 0; ii < IC; ++ii) {
                            const int i = i0 + ii;
                            double a = mq[ii], c = aq[ii];
#pragma unroll
                            for (int m = 0; m < NU; ++m) a -= si[ii][m] * Kj[m], c -= bi[ii][m] * Kj[m];
                            if (lane < NX) {
                                if (i >= lane) lT[i * NX + lane] = a, lT[lane * NX + i] = a;
                                lBA[i * NW + NU + lane] = c;
                            }
                        }
                    }
                }
                {
                    const double *lK = sK();
                    double pk = 0.0;
                    if (lane < NX) {
                        pk = sMV()[NU + lane];
#pragma unroll
                        for (int m = 0; m < NU; ++m) pk -= lK[m * NX + lane] * sMV()[m];
                        p[k * NX + lane] = pk;
                        sPV()[lane] = pk;
                    }
                    wave_sync();
                    d2_t cp[Cfg::NAC2], ca[Cfg::NBA2];
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NAC2; ++s_) {
                        const int e2 = lane + 64 * s_;
                        cp[s_] = *(const d2_t *)(lT + 2 * e2);
                    }
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NBA2; ++s_) {
                        const int e2 = lane + 64 * s_;
                        ca[s_] = *(const d2_t *)(lBA + 2 * e2);
                    }
                    d2_t pp[Cfg::NPK2];
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NPK2; ++s_) pp[s_].x = lT[pko[s_][0]], pp[s_].y = lT[pko[s_][1]];   // past the triangle: words behind the block
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NPK2; ++s_) *(d2_t *)&P[k * Cfg::PST + 2 * (lane + 64 * s_)] = pp[s_];
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NAC2; ++s_) {
                        const int e2 = lane + 64 * s_;
                        if (s_ + 1 < Cfg::NAC2 || 2 * e2 < NW * NW - 1) *(d2_t *)(lP + 2 * e2) = cp[s_];   // M is dead: every lane is past its reads (sync above)
                    }
#pragma unroll
                    for (int s_ = 0; s_ < Cfg::NBA2; ++s_) {
                        const int e2 = lane + 64 * s_;
                        *(d2_t *)&Acl[k * Cfg::BST + 2 * e2] = ca[s_];
                    }
                }
                wave_sync();
                ph(13);
            });
        return ok;
    }

    // ---- backward vector sweep for a new right-hand side g on the stored factors (same bb as the factor sweep: hb is re-used).
    // Produces p_k (workspace), kff_k.
    MPCRL_DI void backward_vec(const WsArr g) {
        constexpr int AST = Cfg::AST, D = Cfg::DEPTH;
        const int li = lane < NX ? lane : 0;
        double pcur = g[N * NW + NU + li];
        if (lane < NX) p[N * NX + lane] = pcur;
        d2_t nA[D][Cfg::NBA2];
        double ngx[D], nhb[D], ngu[D][NU], nK[D][NU];
        double *lA = sA(), *lv = sSV(0);
        staged_loop<D>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                blk_load(Acl + k * Cfg::BST, NX * NW / 2, nA[d], lane);
                ngx[d] = g[k * NW + NU + li], nhb[d] = hb[k * NX + li];
#pragma unroll
                for (int m = 0; m < NU; ++m) ngu[d][m] = g[k * NW + m], nK[d][m] = K[(k * NU + m) * NX + li];
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const double v = pcur + nhb[d];
                if (lane < NX) lv[lane] = v, ccv[k * NX + lane] = v;
                blk_to_lds(lA, NX * NW / 2, nA[d], lane);
                double a = ngx[d];
#pragma unroll
                for (int m = 0; m < NU; ++m) a = fma(-nK[d][m], ngu[d][m], a);   // g_x - K' g_u   (K_0 = 0 in Q-mode)
                refill();
                wave_sync();
                a = lds_dot<NX>(lA + NU + li, NW, lv, a);   // column li of Acl_k (stored where A_k sits in the [B A] block)
                pcur = a;
                if (lane < NX) p[k * NX + lane] = a;
                wave_sync();
            });
        // feed-forward kff_k = R_k^{-1} (g_u + B_k' cc_k), one stage per lane
        for (int k = lane; k < N; k += NT) {
            const bool pin = k == 0 && qmode;
            double mv[NU];
#pragma unroll
            for (int m = 0; m < NU; ++m) mv[m] = g[k * NW + m];
            const WsArr Bk = BA + k * NX * NW;
#pragma unroll 3
            for (int i = 0; i < NX; ++i) {
                const double c = ccv[k * NX + i];
#pragma unroll
                for (int m = 0; m < NU; ++m) mv[m] = fma(Bk[i * NW + m], c, mv[m]);
            }
            const WsArr Lk = L + k * NU * NU;
            double y[NU], z[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = mv[i];
#pragma unroll
                for (int m = 0; m < i; ++m) a -= Lk[i * NU + m] * y[m];
                y[i] = a * Lk[i * NU + i];
            }
#pragma unroll
            for (int i = NU - 1; i >= 0; --i) {
                double a = y[i];
#pragma unroll
                for (int m = i + 1; m < NU; ++m) a -= Lk[m * NU + i] * z[m];
                z[i] = a * Lk[i * NU + i];
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) kff[k * NU + i] = pin ? 0.0 : z[i];
        }
        wave_sync();
    }

    // ---- forward sweep: Dx (serial chain on Acl), then Du and (optionally) Dnu stage-parallel
    template <bool want_nu>
    MPCRL_DI void forward(const WsArr bb) {
        constexpr int AST = Cfg::AST, D = Cfg::DEPTH;
        const int li = lane < NX ? lane : 0;
        double xcur = 0.0;
        if (lane < NX) Dx[lane] = 0.0, Dnu[lane] = 0.0;
        d2_t nA[D][Cfg::NBA2], nP[D][Cfg::NPK2];
        double nbb[D], npv[D], nkf[D][NU];
        // the packed P_k is unpacked (both triangles) while it is published to LDS
        int pij[Cfg::NPK2][2], pji[Cfg::NPK2][2];
        if (want_nu) packed_offsets(pij, false), packed_offsets(pji, true);
        auto p_to_lds = [&](double *dst, const d2_t (&rr)[Cfg::NPK2]) {
#pragma unroll
            for (int s_ = 0; s_ < Cfg::NPK2; ++s_) {
                dst[pij[s_][0]] = rr[s_].x, dst[pji[s_][0]] = rr[s_].x;
                dst[pij[s_][1]] = rr[s_].y, dst[pji[s_][1]] = rr[s_].y;
            }
        };
        double *lA = sA(), *lPk = sPk(), *lv = sSV(0);
        // Lane i takes ROW i of Acl_k here: with the block's own row stride NW (even) a whole lane group would sit on four LDS banks
        // (16-way conflicts on every operand read), so the block is published with the odd stride NW + 1 (as in qp_residuals).
        constexpr int NWP = NW + 1;
        int pdst[Cfg::NBA2][2];
#pragma unroll
        for (int s_ = 0; s_ < Cfg::NBA2; ++s_)
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
                const int e = 2 * (lane + 64 * s_) + h_;
                pdst[s_][h_] = e < NX * NW ? (e / NW) * NWP + e % NW : NX * NWP + (e & 1);   // past the block: a dump slot
            }
        // stage k: dx_{k+1} = Acl_k dx_k + (b_k - B_k kff_k)  (B_k sits beside Acl_k in the staged block);
        // with want_nu also the multiplier step of stage k, Dnu_k = p_k + P_k dx_k (k >= 1)
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
                blk_load(Acl + k * Cfg::BST, NX * NW / 2, nA[d], lane);
                nbb[d] = bb[k * NX + li];
#pragma unroll
                for (int m = 0; m < NU; ++m) nkf[d][m] = kff[k * NU + m];
                if (want_nu) {
                    blk_load(P + k * Cfg::PST, Cfg::PST / 2, nP[d], lane);
                    npv[d] = p[k * NX + li];
                }
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                if (lane < NX) lv[lane] = xcur;
#pragma unroll
                for (int s_ = 0; s_ < Cfg::NBA2; ++s_) lA[pdst[s_][0]] = nA[d][s_].x, lA[pdst[s_][1]] = nA[d][s_].y;
                if (want_nu) p_to_lds(lPk, nP[d]);
                double a = nbb[d], b = want_nu ? npv[d] : 0.0;
                double kf[NU];
#pragma unroll
                for (int m = 0; m < NU; ++m) kf[m] = nkf[d][m];
                refill();
                wave_sync();
                const double *row = lA + li * NWP;
#pragma unroll
                for (int m = 0; m < NU; ++m) a = fma(-row[m], kf[m], a);
                a = lds_dot<NX>(row + NU, 1, lv, a);   // row li of Acl_k
                if (want_nu && k > 0) {
                    b = lds_dot<NX>(lPk + li * NX, 1, lv, b);
                    if (lane < NX) Dnu[k * NX + lane] = b;
                }
                xcur = a;
                if (lane < NX) Dx[(k + 1) * NX + lane] = a;
                wave_sync();
            });
        if (want_nu) {   // terminal multiplier step
            d2_t tP[Cfg::NPK2];
            blk_load(P + N * Cfg::PST, Cfg::PST / 2, tP, lane);
            double b = p[N * NX + li];
            if (lane < NX) lv[lane] = xcur;
            p_to_lds(lPk, tP);
            wave_sync();
            b = lds_dot<NX>(lPk + li * NX, 1, lv, b);
            if (lane < NX) Dnu[N * NX + lane] = b;
        }
        wave_sync();
        for (int e = lane; e < N * NU; e += NT) {
            const int k = e / NU;
            double a = -kff[e];
            const WsArr Kr = K + e * NX;
#pragma unroll 3
            for (int j = 0; j < NX; ++j) a = fma(-Kr[j], Dx[k * NX + j], a);
            Du[e] = a;
        }
        wave_sync();
    }

    // =====================================================================================================================
    // Round-4 sweeps: register-resident MFMA pipelines in Omega coordinates (OmCfg above).  Same contract as factor / backward_vec /
    // forward — workspace in, workspace out (natural-order p, kff, Dx, Du, Dnu for the row phases) — without touching LDS:
    // the round-3 sweeps spent most of a stage waiting for LDS round trips (69 s_waitcnt per factor stage for 49 MFMAs).
    // =====================================================================================================================
    typedef double d4_t __attribute__((ext_vector_type(4)));
    template <int SRC>
    MPCRL_DI static double bcast_lane(double v) {   // the value lane SRC holds, in every lane (two v_readlane_b32)
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, SRC), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), SRC);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    // state index of row slot 4 rg + lr of a register (pad rows of the control group: some valid address, the value is masked)
    template <int RG_>
    MPCRL_DI static int om_xr(int lr) { return 4 * RG_ + lr - (RG_ >= OmCfg<M>::GQ ? NU : 0); }
    // stage-vector index [u; x] of slot 4 rg + lr
    template <int RG_>
    MPCRL_DI static int om_nat(int lr) {
        using O = OmCfg<M>;
        if constexpr (RG_ < O::GQ) return NU + 4 * RG_ + lr;
        if constexpr (RG_ > O::GQ) return 4 * RG_ + lr;
        return lr < NU ? lr : O::Q + lr;
    }

    // ---- factor sweep (see OmCfg): P_{k+1} stays in registers in the result layout, which is its operand layout for T = P W.
    // slot of entry i of the stage vector [u; x]
    MPCRL_DI static int om_slot(int i) { return i < NU ? OmCfg<M>::Q + i : (i - NU < OmCfg<M>::Q ? i - NU : i); }

    // ---- y_k = [B A]_k' nu_{k+1}, k = 0 .. N - 1, for a multiplier array nu [(N+1) NX]: into ly (LDS, [k HBS + slot of the stage
    // vector]); lnu (LDS) receives nu in Omega order.  Stage-parallel (no chain): [B A]_k as it lies in the workspace is the A
    // operand, the vector the B operand, 4 stages of operands in flight.
    MPCRL_DI void wt_nu_pass(const WsArr nu, double *lnu, double *ly) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        stage_vec_lds<true>(lnu, nu, N);
        wave_sync();
        int colnat[NTR];
#pragma unroll
        for (int tj = 0; tj < NTR; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const int rbase = lr * NW;
        double nA[D][RG][NTR];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) nA[d][rg][tj] = BA[k * NX * NW + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW];
                });
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                double Ak[RG][NTR], vop[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) {
                        double v = nA[d][rg][tj];
                        if constexpr (rg == GQ) v = padl ? 0.0 : v;
                        if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                        Ak[rg][tj] = v;
                    }
                    vop[rg] = lnu[(k + 1) * O::HBS + 4 * rg + lr];
                });
                refill();
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[ks][ti], vop[ks], acc[ti], 0, 0, 0);
                if (lc == 0)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        ly[k * O::HBS + 4 * rg + lr] = acc[rg / 4][rg % 4];
                    });
            });
        wave_sync();
    }

    // ---- residuals of the QP at its START (what qp_solve asks for with the residual scaling on: later iterations scale them):
    // there the direction of every stage but the first is zero — dx_0 = x0 - X_0 and a pinned du_0 are the only entries of (dx, du)
    // the QP starts with — so  rb_k = r_k,  rg_k = q_k -+ lam + [B A]_k' nuq_{k+1} - [0; nuq_k], plus [B A]_0 dv_0 and c_0 H dv_0 at
    // stage 0 where dv_0 != 0 (first QP of a warm solve from a new initial state).  One MFMA pass + one pass over the entries,
    // against a stage-serial pass with [B A]_k staged through LDS (round 3: 7 % / 14 % of the kernel at n_mass 5 / 7).
    MPCRL_DI double qp_residuals2() {
        using O = OmCfg<M>;
        const int ne = (N + 1) * NW;
        double *const lnu = lds + Cfg::oBig, *const ly = lnu + (N + 1) * O::HBS;
        wt_nu_pass(nuq, lnu, ly);
        double rloc = 0.0;
        batched_pass<4>(ne, lane,
                        [&](int e) {
                            const int k = e / NW, i = e - k * NW;
                            return Quad4{q[e], lam[e], lam[ne + e], (i >= NU && k > 0) ? nuq[k * NX + i - NU] : 0.0};
                        },
                        [&](int e, const Quad4 &v) {
                            const int k = e / NW, i = e - k * NW;
                            double g = v.a;
                            if (!skipc(k, i)) {
                                if (has(0, k, i)) g -= v.b;
                                if (has(1, k, i)) g += v.c;
                            }
                            g += (k < N ? ly[k * O::HBS + om_slot(i)] : 0.0) - v.d;
                            if (fixedc(k, i) || skipc(k, i)) g = 0.0;
                            rg[e] = g, rloc = fmax(rloc, fabs(g));
                        });
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int e, double v) { rb[e] = v, rloc = fmax(rloc, fabs(v)); });
        wave_sync();
        rloc = fmax(rloc, stage0_direction_terms());
        return rloc;
    }

    // ---- the same at no pass over the [B A]_k: round_start evaluated rg at the multipliers the QP starts from (qp_solve, rg_ready)
    MPCRL_DI double qp_start_residuals() {
        const int ne = (N + 1) * NW;
        double rloc = 0.0;
        batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int, double v) { rloc = fmax(rloc, fabs(v)); });
        batched_pass<8>(N * NX, lane, [&](int e) { return r[e]; }, [&](int e, double v) { rb[e] = v, rloc = fmax(rloc, fabs(v)); });
        wave_sync();
        return fmax(rloc, stage0_direction_terms());
    }
    // stage 0 with a non-zero direction (dx_0 = x0 - X_0 of a warm solve from a new state, a pinned du_0): [B A]_0 dv_0 into rb_0,
    // c_0 H dv_0 into rg_0; one row / column per lane.  Returns the largest entry it changed.
    MPCRL_DI double stage0_direction_terms() {
        double rloc = 0.0, d0 = 0.0;
        if (lane < NX) d0 = fabs(dx[lane]);
        if (lane < NU) d0 = fmax(d0, fabs(du[lane]));
        if (wave_max(d0) > 0.0) {
            double dv[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) dv[j] = j < NU ? du[j < NU ? j : 0] : dx[j >= NU ? j - NU : 0];
            if (lane < NX) {
                double a = rb[lane];
#pragma unroll 4
                for (int j = 0; j < NW; ++j) a = fma(BA[lane * NW + j], dv[j], a);
                rb[lane] = a, rloc = fmax(rloc, fabs(a));
            }
            if (lane < NW && !fixedc(0, lane)) {
                double hd = 0.0;
#pragma unroll 4
                for (int j = 0; j < NW; ++j) hd = fma(M::hess(false, lane < NW ? lane : 0, j, th), dv[j], hd);
                const double g = fma(ck(0), hd, rg[lane]);
                rg[lane] = g, rloc = fmax(rloc, fabs(g));
            }
            wave_sync();
        }
        return rloc;
    }

    // STORE_P: P_k goes to HBM as well (only the adjoint solves of the sensitivities multiply with it afterwards: forward2<true>).
    template <class HS, bool STORE_P>
    MPCRL_DI bool factor2(HS &hs, const WsArr g, const WsArr bb) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NT = O::NT, NTR = O::NTR, GQ = O::GQ, TQ = O::TQ, RQ = O::RQ, LQ = O::LQ, TV = O::TV, LV = O::LV;
        constexpr int FD = NTR <= 2 ? 2 : 1;     // prefetch slots (a stage is 3 - 9 us of work: one stage ahead covers the HBM latency; registers at n_mass 7)
        constexpr bool RAGGED = 4 * RG > NW;     // the last row group runs past NW
        const int lr = lane >> 4, lc = lane & 15;
        hs.begin(lane);
        bool ok = true;
        int colnat[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const bool vcl = lc == LV, wcl = lc < LV, padl = lr < NU, ucl = lc >= LQ && lc < LQ + NU;
        const double vcm = vcl ? 1.0 : 0.0;
        const int rbase = lr * NW;
        int btoff[NTR];
        bool btok[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int e = 16 * ti + lc, xr = O::xrow(e < NW ? e : 0);
            btok[ti] = e < NW && xr >= 0 && lr < NU;
            btoff[ti] = (btok[ti] ? xr : 0) * NW + (lr < NU ? lr : 0);
        }
        const unsigned gbase = (unsigned)lane, cbase = (unsigned)(lr * LV + (lc < LV ? lc : 0));
        // ---- terminal stage: P_N = c_N hess l_N + D_N, p_N = g_N
        d4_t Pt[NTR][NT];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    double v = 0.0;
                    if constexpr (rg < RG) {
                        const int e = 4 * rg + lr, c = 16 * tj + lc;
                        const int xr = O::xrow(e < NW ? e : 0), xc = O::xrow(c < NW ? c : 0);
                        const bool rok = e < NW && xr >= 0;
                        if (rok && c < NW && xc >= 0) {
                            v = hs.term(N, xr > xc ? xr : xc, xr > xc ? xc : xr);
                            if (xr == xc) v += Dg[N * NW + NU + xr];
                        } else if (rok && c == O::VC)
                            v = g[N * NW + NU + xr];
                        if (STORE_P && (tj < TV || lc < LV)) P2[N * O::GSZ + O::goff(rg, tj, lr, lc)] = v;
                        if (rok && c == O::VC) p[N * NX + xr] = v;
                    }
                    Pt[ti][tj][r] = v;
                }
            });
        });
        // per stage and lane: W (RG x NT registers; in the tile of the vector column the lane of that column fetches b instead),
        // B' (NTR), and ONE register per row group for the right-hand side and the barrier diagonal (the lane of the vector column
        // fetches g, the diagonal lane D: they are different lanes for every valid row)
        // (one lane is both: the diagonal lane of row 16 ti + LV in a tile ti != TV is the lane of the vector column — its D comes
        // with a load of its own, ndx)
        double nW[FD][RG][NT], nBt[FD][NTR], ngd[FD][RG], ndx[FD][NTR];
        const unsigned bbrel = bb.off - BA.off, grel = g.off - BA.off, dgrel = Dg.off - BA.off;
        hs.prefetch(N - 1);
        staged_loop<FD>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const WsArr Bk = BA + k * NX * NW;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const int xr = om_xr<rg>(lr), nt = om_nat<rg>(lr);
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        const int ow = k * NX * NW + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW;
                        nW[d][rg][tj] = BA[(tj == TV && vcl) ? (int)bbrel + k * NX + xr : ow];
                    }
                    const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                    ngd[d][rg] = BA[(int)(vcl ? grel : dgrel) + k * NW + (rok ? nt : 0)];
                });
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) {
                    nBt[d][ti] = Bk[btoff[ti]];
                    ndx[d][ti] = (ti != TV && 16 * ti + LV < NW) ? Dg[k * NW + O::nat(16 * ti + LV < NW ? 16 * ti + LV : 0)] : 0.0;
                }
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                const bool pin = k == 0 && qmode;
                hs.advance(k);
                // ---- the stage operands out of their prefetch slot: W = [A B | b] with its pad rows, B' as an A operand
                d4_t Wt[NTR][NT];
                double gd[RG], Bt[NTR], dgx[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) dgx[ti] = ndx[d][ti];
                static_for<NTR>([&](auto ti_) {
                    static_for<4>([&](auto r_) {
                        constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) {
                            double v = 0.0;
                            if constexpr (rg < RG) {
                                v = nW[d][rg][tj];
                                if (tj == TV) v = lc <= LV ? v : 0.0;
                                if constexpr (rg == GQ) v = padl ? 0.0 : v;
                                if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                            }
                            Wt[ti][tj][r] = v;
                        }
                        if constexpr (rg < RG) {
                            const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                            gd[rg] = rok ? ngd[d][rg] : 0.0;
                        }
                    });
                });
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) Bt[ti] = btok[ti] ? nBt[d][ti] : 0.0;
                refill();
                // ---- T = P W, row tile by row tile (the column tile of P it read is dead afterwards)
                d4_t Tt[NTR][NT];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[ks / 4][ti][ks % 4], Wt[ks / 4][tj][ks % 4], acc, 0, 0, 0);
                        Tt[ti][tj] = acc;
                    }
                ph(10);
                // its vector column is P b (kept: hb, stored below), then + p
                double hbv[RG];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * ti + r < RG) {
                            hbv[4 * ti + r] = Tt[ti][TV][r];
                            Tt[ti][TV][r] = fma(Pt[ti][TV][r], vcm, Tt[ti][TV][r]);
                        }
                // ---- M = H + D + W' T, vector column g + W' (P b + p); column tile by column tile (T's is dead afterwards).  The column
                // tile of the control group goes first: its diagonal tile feeds the Cholesky below, whose serial VALU chain (~1.5 k cycles)
                // then runs in the shadow of the other column tiles' MFMAs
                d4_t Mt[NTR][NT];
                auto m_column = [&](auto tj_) {
                    constexpr int tj = decltype(tj_)::value;
                    static_for<NTR>([&](auto tmo_) {
                        constexpr int tmo = decltype(tmo_)::value, tm = tmo == 0 ? TQ : (tmo <= TQ ? tmo - 1 : tmo);     // row tile TQ first
                        d4_t acc;
                        static_for<4>([&](auto r_) {
                            constexpr int r = decltype(r_)::value, rg = 4 * tm + r;
                            double c = 0.0;
                            if constexpr (rg < RG) {
                                c = hs.tile(k, rg, tj);
                                if (tj == tm) c += (lc == 4 * r + lr) ? ((tm != TV && r == LV / 4 && vcl) ? dgx[tm] : gd[rg]) : 0.0;
                                if (tj == TV) c += vcl ? gd[rg] : 0.0;
                            }
                            acc[r] = c;
                        });
#pragma unroll
                        for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[ks / 4][tm][ks % 4], Tt[ks / 4][tj][ks % 4], acc, 0, 0, 0);
                        Mt[tm][tj] = acc;
                    });
                };
                m_column(std::integral_constant<int, TQ>{});
                // ---- Cholesky of the control block on broadcast values (every lane, redundantly), then its inverse
                double Minv[NU][NU];
                {
                    const double src = Mt[TQ][TQ][RQ];
                    double a_[NU][NU], Lc[NU][NU], Li[NU][NU];
                    static_for<NU>([&](auto i_) {
                        static_for<NU>([&](auto j_) {
                            constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
                            if constexpr (j <= i) a_[i][j] = bcast_lane<16 * i + LQ + j>(src);
                        });
                    });
                    bool okc = true;
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double a = a_[i][j];
#pragma unroll
                            for (int m = 0; m < j; ++m) a -= Lc[i][m] * Lc[j][m];
                            if (i == j) {
                                okc = okc && (a > 0.0);
                                // 1 / sqrt(a) from the hardware seed + two Newton steps (~1 ulp): the pivot is a positive normal number
                                // wherever the result is used, and an IEEE sqrt followed by an IEEE division is ~40 dependent
                                // instructions on the critical path of every stage
                                double y = __builtin_amdgcn_rsq(a);
                                y = y * fma(-0.5 * a * y, y, 1.5);
                                Lc[i][i] = y * fma(-0.5 * a * y, y, 1.5);
                            } else
                                Lc[i][j] = a * Lc[j][j];
                        }
                    ok = ok && (okc || pin);
                    // Li = L^-1 (lower; Lc carries the inverted diagonal), Minv = Li' Li
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        Li[j][j] = Lc[j][j];
#pragma unroll
                        for (int i = j + 1; i < NU; ++i) {
                            double a = 0.0;
#pragma unroll
                            for (int m = j; m < i; ++m) a -= Lc[i][m] * Li[m][j];
                            Li[i][j] = a * Lc[i][i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NU; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double a = 0.0;
#pragma unroll
                            for (int m = i; m < NU; ++m) a = fma(Li[m][i], Li[m][j], a);
                            Minv[i][j] = a, Minv[j][i] = a;
                        }
                }
                static_for<NT>([&](auto tj_) {
                    if constexpr (decltype(tj_)::value != TQ) m_column(tj_);
                });
                ph(11);
                double minvop = 0.0;     // A operand of K = R^-1 [S | R | mv_u]: R^-1(i, l) at lane (lr = l, lc = i)
#pragma unroll
                for (int i = 0; i < NU; ++i)
#pragma unroll
                    for (int l = 0; l < NU; ++l) minvop = (lc == i && lr == l) ? Minv[i][l] : minvop;
                if (pin) minvop = 0.0;
                ph(12);
                // ---- K (register 0 of one MFMA per column tile), the rank-NU update P' = M - S' K, G = W - B K with -K in the pad rows
                double nK[NT], nKz[NT], Sr[NTR];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const d4_t z4 = {0.0, 0.0, 0.0, 0.0};
                    const d4_t kt = __builtin_amdgcn_mfma_f64_16x16x4f64(minvop, Mt[TQ][tj][RQ], z4, 0, 0, 0);
                    nK[tj] = -kt[0];
                    nKz[tj] = (tj == TQ && ucl) ? 0.0 : nK[tj];
                }
#pragma unroll
                for (int ta = 0; ta < NTR; ++ta) Sr[ta] = Mt[TQ][ta][RQ];
#pragma unroll
                for (int ta = 0; ta < NTR; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb) Mt[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Sr[ta], nK[tb], Mt[ta][tb], 0, 0, 0);
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) Wt[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bt[ti], nKz[tj], Wt[ti][tj], 0, 0, 0);
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) Wt[TQ][tj][RQ] = padl ? nKz[tj] : Wt[TQ][tj][RQ];
                // ---- out: G_k and P_k in the register layout (full 512-byte bursts); from the lanes of the vector column hb_k, and
                // p_k, kff_k in natural order for the other phases; R^-1 from the lanes that hold its entries
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < TV; ++tj) {
                        G2[k * O::GSZ + (rg * TV + tj) * 64 + gbase] = Wt[rg / 4][tj][rg % 4];
                        if constexpr (STORE_P) P2[k * O::GSZ + (rg * TV + tj) * 64 + gbase] = Mt[rg / 4][tj][rg % 4];
                    }
                });
                if (wcl)      // the last column tile: its columns < NW, compactly
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        G2[k * O::GSZ + RG * TV * 64 + rg * O::CT + cbase] = Wt[rg / 4][TV][rg % 4];
                        if constexpr (STORE_P) P2[k * O::GSZ + RG * TV * 64 + rg * O::CT + cbase] = Mt[rg / 4][TV][rg % 4];
                    });
                if (vcl) {
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        hb2[k * O::HBS + 4 * rg + lr] = hbv[rg];
                        // natural p: the pad rows (and rows past NW) go to a dump slot behind the array of stage N
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        p[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = Mt[rg / 4][TV][rg % 4];
                    });
                    kff[padl ? k * NU + lr : N * NU] = -nK[TV];
                }
                if (lc < NU && lr < NU) minv2[k * 16 + 4 * lc + lr] = minvop;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) Pt[ti][tj] = Mt[ti][tj];
                ph(13);
            });
        return ok;
    }

    // One vector of the whole horizon from the workspace into LDS in Omega order, dst[k HBS + slot]: natural stage-vector arrays
    // (stride NW: slot -> nat(slot)) or state arrays (stride NX: slot -> state index, control slots 0); kmax = last stage.
    template <bool STATE>
    MPCRL_DI void stage_vec_lds(double *dst, const WsArr src, int kmax) {
        using O = OmCfg<M>;
        const int n = (kmax + 1) * O::HBS;
        batched_pass<4>(n, lane,
                        [&](int e) {
                            const int k = e / O::HBS, sl = e - k * O::HBS;
                            if (STATE) {
                                const int xr = O::xrow(sl < NW ? sl : 0);
                                const double v = src[k * NX + (sl < NW && xr >= 0 ? xr : 0)];
                                return (sl < NW && xr >= 0) ? v : 0.0;
                            } else {
                                const double v = src[k * NW + (sl < NW ? O::nat(sl) : 0)];
                                return sl < NW ? v : 0.0;
                            }
                        },
                        [&](int e, double v) { dst[e] = v; });
    }

    // ---- backward vector sweep for a new right-hand side g on the stored G_k: [p_k; mv_u] = g + G_k' [p_{k+1} + hb_k; g_u],
    // a chain of MFMAs whose B operand is the previous result (column 0 of the lanes carries the vector), then kff = R^-1 mv_u.
    // The two vectors of the horizon (g, hb) are staged in LDS in Omega order first: the stream of the G_k blocks is all that is
    // left in the global-memory queue (the sweep is bound by HBM bandwidth: depth x block = bytes in flight).
    MPCRL_DI void backward_vec2(const WsArr g) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, TV = O::TV, D = NTR <= 2 ? 4 : 2;   // (stages in flight: registers at n_mass 7)
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        double *const lg = lds + Cfg::oBig, *const lhb = lg + (N + 1) * O::HBS;
        stage_vec_lds<false>(lg, g, N);
        batched_pass<4>(N * O::HBS, lane, [&](int e) { return hb2[e]; }, [&](int e, double v) { lhb[e] = v; });
        wave_sync();
        unsigned goffs[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) goffs[ti] = O::goff(0, ti, lr, lc);
        d4_t R[NTR];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                double v = 0.0;
                if constexpr (rg < RG) {
                    v = lg[N * O::HBS + 4 * rg + lr];
                    if constexpr (rg == GQ) v = padl ? 0.0 : v;
                }
                R[ti][r] = v;
            });
        });
        if (lc == 0)
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                p[rok ? N * NX + om_xr<rg>(lr) : (N + 1) * NX] = R[rg / 4][rg % 4];
            });
        double nG[D][RG][NTR];
        staged_loop<D>(
            N,
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) nG[d][rg][ti] = G2[k * O::GSZ + goffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
                });
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                double vop[RG], gt[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    gt[rg] = lg[k * O::HBS + 4 * rg + lr];
                    double v = R[rg / 4][rg % 4] + lhb[k * O::HBS + 4 * rg + lr];
                    if constexpr (rg == GQ) v = padl ? gt[rg] : v;
                    vop[rg] = v;
                });
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ti][r] = 4 * ti + r < RG ? gt[4 * ti + r < RG ? 4 * ti + r : 0] : 0.0;
                double Gk[RG][NTR];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) Gk[rg][ti] = nG[d][rg][ti];
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ks][ti], vop[ks], acc[ti], 0, 0, 0);
                if (lc == 0)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        p[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = v;
                        if constexpr (rg == GQ) mvu2[k * 4 + lr] = v;      // (lane lr = 3 of the group: a state row, never read)
                    });
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) R[ti] = acc[ti];
            });
        wave_sync();
        // feed-forward kff_k = R_k^-1 mv_u, one stage per lane
        for (int k = lane; k < N; k += NT) {
            double mv[NU];
#pragma unroll
            for (int m = 0; m < NU; ++m) mv[m] = mvu2[k * 4 + m];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < NU; ++m) a = fma(minv2[k * 16 + 4 * i + m], mv[m], a);
                kff[k * NU + i] = a;     // (R^-1 is stored as zero at a pinned stage 0)
            }
        }
        wave_sync();
    }

    // ---- forward sweep: [dx_{k+1}; du_k] = [b_k; -kff_k] + G_k [dx_k; -kff_k] (the control slots of the operand carry -kff: the
    // x rows of G hold [Acl | B], its pad rows [-K | 0]), with want_nu also Dnu_k = p_k + P_k dx_k on the same operand.
    // G_k is wanted as an A operand (contraction over its COLUMNS): the transposed access pattern of the streamed block.
    template <bool want_nu>
    MPCRL_DI void forward2(const WsArr bb) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, TV = O::TV, LV = O::LV, D = NTR <= 2 ? (want_nu ? 3 : 4) : (want_nu ? 1 : 2);
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        // staged in LDS, Omega order: [b_k; kff_k] (control slots: kff), with want_nu p_k
        double *const lbk = lds + Cfg::oBig, *const lp = lbk + (N + 1) * O::HBS;
        stage_vec_lds<true>(lbk, bb, N - 1);
        if (want_nu) stage_vec_lds<true>(lp, p, N);
        wave_sync();
        for (int e = lane; e < N * NU; e += NT) {
            const int k = e / NU;
            lbk[k * O::HBS + O::Q + (e - k * NU)] = kff[e];
        }
        wave_sync();
        // element (row a, column b) of a streamed block: this lane wants a = 16 ti + lc (rows past the block: its last row), b = 4 ks + lr
        unsigned tfull[NTR], tcomp[NTR], poffs[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int a_ = 16 * ti + lc, a = a_ < 4 * RG ? a_ : 4 * RG - 1;
            tfull[ti] = (unsigned)((a >> 2) * TV * 64 + (a & 3) * 16 + lr);
            tcomp[ti] = (unsigned)(RG * TV * 64 + (a >> 2) * O::CT + (a & 3) * LV + lr);
            poffs[ti] = O::goff(0, ti, lr, lc);
        }
        if (lane < NX) Dx[lane] = 0.0, Dnu[lane] = 0.0;
        double w[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) w[rg] = 0.0;
        double nGt[D][NTR][RG], nP[D][want_nu ? RG : 1][want_nu ? NTR : 1];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks)
                        nGt[d][ti][ks] = G2[k * O::GSZ + (ks / 4 < TV ? tfull[ti] + (ks / 4) * 64 : tcomp[ti]) + 4 * (ks % 4)];
                if constexpr (want_nu)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
#pragma unroll
                        for (int ti = 0; ti < NTR; ++ti) nP[d][rg][ti] = P2[k * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
                    });
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                d4_t acc[NTR], acc2[NTR];
                static_for<NTR>([&](auto ti_) {
                    static_for<4>([&](auto r_) {
                        constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                        double c = 0.0, c2 = 0.0;
                        if constexpr (rg < RG) {
                            c = lbk[k * O::HBS + 4 * rg + lr];
                            if constexpr (rg == GQ) c = padl ? -c : c;       // -kff in the control slots
                            if constexpr (want_nu) c2 = lp[k * O::HBS + 4 * rg + lr];
                        }
                        acc[ti][r] = c, acc2[ti][r] = c2;
                    });
                });
                w[GQ] = padl ? acc[GQ / 4][GQ % 4] : w[GQ];                   // operand: -kff_k in the control slots as well
                double Gk[NTR][RG], Pk[want_nu ? RG : 1][want_nu ? NTR : 1];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks) Gk[ti][ks] = nGt[d][ti][ks];
                if constexpr (want_nu) {
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                        for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = nP[d][rg][ti];
                }
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) {
                        acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ti][ks], w[ks], acc[ti], 0, 0, 0);
                        if constexpr (want_nu) acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
                    }
                if (lc == 0)      // the vector sits in column 0: rows past NW and (for Dnu) the control slots go to the dump slots
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        if constexpr (rg == GQ)
                            Dx[padl ? (int)(Du.off - Dx.off) + k * NU + lr : (k + 1) * NX + om_xr<rg>(lr)] = v;     // du_k / dx_{k+1}
                        else
                            Dx[rok ? (k + 1) * NX + om_xr<rg>(lr) : (N + 1) * NX] = v;
                        if constexpr (want_nu) Dnu[(rok && k > 0) ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = acc2[rg / 4][rg % 4];
                    });
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) w[rg] = acc[rg / 4][rg % 4];
            });
        if constexpr (want_nu) {   // terminal multiplier step: Dnu_N = p_N + P_N dx_N
            d4_t acc2[NTR];
            double Pk[RG][NTR];
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = P2[N * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
                acc2[rg / 4][rg % 4] = lp[N * O::HBS + 4 * rg + lr];
            });
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * ti + r >= RG) acc2[ti][r] = 0.0;
            w[GQ] = padl ? 0.0 : w[GQ];
#pragma unroll
            for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
            if (lc == 0)
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                    Dnu[rok ? N * NX + om_xr<rg>(lr) : (N + 1) * NX] = acc2[rg / 4][rg % 4];
                });
        }
        wave_sync();
    }

    // ---- the NU adjoint solves of the sensitivities in ONE forward sweep: right-hand side -e_iu in the controls of stage 0, no
    // dynamics offset, so p_k = 0 and kff_k = 0 for k >= 1 and kff_0 = -R_0^-1 e_iu; solve iu rides in column iu of the B operand
    // (the 16 columns of the MFMA cost the same as one).  Out: Ydx / Ydu / Ydnu [iu][...] as chain_sens_mix / chain_sens_out read them.
    MPCRL_DI void forward2_sens(const WsArr Ydx, const WsArr Ydu, const WsArr Ydnu) {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NTR = O::NTR, GQ = O::GQ, TV = O::TV, LV = O::LV, D = NTR <= 2 ? 3 : 1;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU, col = lc < NU;
        const int sx = (N + 1) * NX, su = N * NU, cj = col ? lc : 0;
        unsigned tfull[NTR], tcomp[NTR], poffs[NTR];
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti) {
            const int a_ = 16 * ti + lc, a = a_ < 4 * RG ? a_ : 4 * RG - 1;
            tfull[ti] = (unsigned)((a >> 2) * TV * 64 + (a & 3) * 16 + lr);
            tcomp[ti] = (unsigned)(RG * TV * 64 + (a >> 2) * O::CT + (a & 3) * LV + lr);
            poffs[ti] = O::goff(0, ti, lr, lc);
        }
        for (int e = lane; e < NU * NX; e += NT) {
            const int j = e / NX;
            Ydx[j * sx + (e - j * NX)] = 0.0, Ydnu[j * sx + (e - j * NX)] = 0.0;
        }
        // -kff_0 of solve lc: column lc of R_0^-1, in the control slots
        const double k0 = (padl && col) ? minv2[4 * lr + cj] : 0.0;
        double w[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) w[rg] = 0.0;
        w[GQ] = k0;
        double nGt[D][NTR][RG], nP[D][RG][NTR];
        staged_loop<D>(
            N,
            [&](int k, auto sl) {
                constexpr int d = decltype(sl)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks)
                        nGt[d][ti][ks] = G2[k * O::GSZ + (ks / 4 < TV ? tfull[ti] + (ks / 4) * 64 : tcomp[ti]) + 4 * (ks % 4)];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) nP[d][rg][ti] = P2[k * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
                });
            },
            [&](int k, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                d4_t acc[NTR], acc2[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc[ti] = d4_t{0.0, 0.0, 0.0, 0.0}, acc2[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
                acc[GQ / 4][GQ % 4] = k == 0 ? k0 : 0.0;          // [b; -kff]: only -kff_0
                w[GQ] = padl ? acc[GQ / 4][GQ % 4] : w[GQ];
                double Gk[NTR][RG], Pk[RG][NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int ks = 0; ks < RG; ++ks) Gk[ti][ks] = nGt[d][ti][ks];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = nP[d][rg][ti];
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) {
                        acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gk[ti][ks], w[ks], acc[ti], 0, 0, 0);
                        acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
                    }
                if (col)
                    static_for<RG>([&](auto rg_) {
                        constexpr int rg = decltype(rg_)::value;
                        const double v = acc[rg / 4][rg % 4];
                        const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                        if constexpr (rg == GQ)
                            Ydx[padl ? (int)(Ydu.off - Ydx.off) + cj * su + k * NU + lr : cj * sx + (k + 1) * NX + om_xr<rg>(lr)] = v;
                        else
                            Ydx[rok ? cj * sx + (k + 1) * NX + om_xr<rg>(lr) : NU * sx] = v;
                        Ydnu[(rok && k > 0) ? cj * sx + k * NX + om_xr<rg>(lr) : NU * sx] = acc2[rg / 4][rg % 4];
                    });
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) w[rg] = acc[rg / 4][rg % 4];
            });
        {   // terminal multiplier step: Dnu_N = P_N dx_N
            d4_t acc2[NTR];
            double Pk[RG][NTR];
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) Pk[rg][ti] = P2[N * O::GSZ + poffs[ti] + rg * (ti < TV ? TV * 64 : O::CT)];
            });
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti) acc2[ti] = d4_t{0.0, 0.0, 0.0, 0.0};
            w[GQ] = padl ? 0.0 : w[GQ];
#pragma unroll
            for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) acc2[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[ks][ti], w[ks], acc2[ti], 0, 0, 0);
            if (col)
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                    Ydnu[rok ? cj * sx + N * NX + om_xr<rg>(lr) : NU * sx] = acc2[rg / 4][rg % 4];
                });
        }
        wave_sync();
    }

    // ---- multipliers of the dynamics at the end of a QP, by ONE costate sweep instead of Dnu_k = p_k + P_k Dx_k in every
    // interior-point iteration (which made the factor sweep stream P_k out and the corrector's forward sweep stream it back in: 9 KB
    // of the 32 KB a stage moved per iteration at n_mass 5).  No step of the iteration uses nuq — the Riccati direction gives Dx, Du,
    // and the bound rows give the step length — it is only an output, and the x rows of the QP's stationarity residual tie it to
    // what the iteration does carry:
    //     rg_x,k = q_x,k + (H dv_k)_x + A_k' nuq_{k+1} - nuq_k -+ lam_x,k        (rg: kept current by the (1 - alpha) scaling)
    // so  nuq_k = [q + H dv -+ lam - rg]_x,k + A_k' nuq_{k+1},  nuq_N = [..]_x,N : the same numbers as the accumulated steps, to
    // rounding.  H dv for all stages is three batches of MFMAs (16 stages per batch as the 16 columns of the B operand), the sweep a
    // chain of MFMAs on [B A]_k as it lies in the workspace.
    MPCRL_DI void costate_nu() {
        using O = OmCfg<M>;
        constexpr int RG = O::RG, NT_ = O::NT, NTR = O::NTR, GQ = O::GQ, D = NTR <= 2 ? 4 : 2;
        constexpr bool RAGGED = 4 * RG > NW;
        const int lr = lane >> 4, lc = lane & 15;
        const bool padl = lr < NU;
        const int ne = (N + 1) * NW;
        double *const lc_ = lds + Cfg::oBig, *const ltab = lc_ + (N + 1) * O::HBS;      // the stage vectors c_k (Omega order), the Hessian table
        // Hessian table in the register layout (= its A-operand layout: H is symmetric)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int tj = 0; tj < NTR; ++tj) {
                const int e = 4 * rg + lr, c = 16 * tj + lc;
                const bool in = e < NW && c < NW;
                ltab[(rg * NTR + tj) * 64 + lane] = in ? M::hess(false, O::nat(in ? e : 0), O::nat(in ? c : 0), th) : 0.0;
            }
        wave_sync();
        for (int b0 = 0; b0 <= N; b0 += 16) {
            const int st = b0 + lc, stc = st <= N ? st : N;      // this lane's stage (column lc of the batch)
            double op[RG];
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = !RAGGED || rg < RG - 1 || 4 * rg + lr < NW;
                double v;
                if constexpr (rg == GQ) {      // the control slots of the group take du (none at the terminal stage)
                    const double t_ = WsArr{dx.base, padl ? du.off + (unsigned)((stc < N ? stc : N - 1) * NU + lr) : dx.off + (unsigned)(stc * NX + om_xr<rg>(lr))}[0];
                    v = (padl && stc >= N) ? 0.0 : t_;
                } else
                    v = dx[stc * NX + (rok ? om_xr<rg>(lr) : 0)];
                op[rg] = rok ? v : 0.0;
            });
            d4_t y[NTR];
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti) {
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < RG; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ltab[(ks * NTR + ti) * 64 + lane], op[ks], acc, 0, 0, 0);
                y[ti] = acc;
            }
            const double cks = ck(stc);
            static_for<RG>([&](auto rg_) {
                constexpr int rg = decltype(rg_)::value;
                const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                const int e = stc * NW + (rok ? om_nat<rg>(lr) : 0), i = rok ? om_nat<rg>(lr) : NU;
                double v = fma(cks, y[rg / 4][rg % 4], q[e]) - this->rg[e];
                if (has(0, stc, i)) v -= lam[e];
                if (has(1, stc, i)) v += lam[ne + e];
                if (st <= N) lc_[st * O::HBS + 4 * rg + lr] = rok ? v : 0.0;
            });
        }
        wave_sync();
        // the chain: nuq_k = c_k + A_k' nuq_{k+1}  (operand rows: next state, pad rows 0; columns: the state slots of stage k)
        int colnat[NTR];
#pragma unroll
        for (int tj = 0; tj < NTR; ++tj) {
            const int c = 16 * tj + lc;
            colnat[tj] = c < NW ? O::nat(c < NW ? c : 0) : 0;
        }
        const int rbase = lr * NW;
        d4_t R[NTR];
        static_for<NTR>([&](auto ti_) {
            static_for<4>([&](auto r_) {
                constexpr int ti = decltype(ti_)::value, r = decltype(r_)::value, rg = 4 * ti + r;
                double v = 0.0;
                if constexpr (rg < RG) v = lc_[N * O::HBS + 4 * rg + lr];
                R[ti][r] = v;
            });
        });
        auto store_nu = [&](int k) {
            if (lc == 0)
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
                    const bool rok = (rg != GQ || !padl) && (!RAGGED || rg < RG - 1 || 4 * rg + lr < NW);
                    nuq[rok ? k * NX + om_xr<rg>(lr) : (N + 1) * NX] = R[rg / 4][rg % 4];
                });
        };
        store_nu(N);
        double nA[D][RG][NTR];
        staged_loop<D>(
            N - 1,     // stages N - 1 .. 1 (the multiplier of the initial condition is not an iterate)
            [&](int idx, auto sl) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) nA[d][rg][tj] = BA[k * NX * NW + rbase + colnat[tj] + (4 * rg - (rg >= GQ ? NU : 0)) * NW];
                });
            },
            [&](int idx, auto sl, auto refill) {
                constexpr int d = decltype(sl)::value;
                const int k = N - 1 - idx;
                double Ak[RG][NTR], vop[RG];
                static_for<RG>([&](auto rg_) {
                    constexpr int rg = decltype(rg_)::value;
#pragma unroll
                    for (int tj = 0; tj < NTR; ++tj) {
                        double v = nA[d][rg][tj];
                        if constexpr (rg == GQ) v = padl ? 0.0 : v;
                        if constexpr (RAGGED && rg == RG - 1) v = 4 * rg + lr < NW ? v : 0.0;
                        Ak[rg][tj] = v;
                    }
                    vop[rg] = R[rg / 4][rg % 4];
                });
                d4_t acc[NTR];
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ti][r] = 4 * ti + r < RG ? lc_[k * O::HBS + 4 * (4 * ti + r < RG ? 4 * ti + r : 0) + lr] : 0.0;
                refill();
#pragma unroll
                for (int ks = 0; ks < RG; ++ks)
#pragma unroll
                    for (int ti = 0; ti < NTR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[ks][ti], vop[ks], acc[ti], 0, 0, 0);
#pragma unroll
                for (int ti = 0; ti < NTR; ++ti) R[ti] = acc[ti];
                // the control slots of the result (B' nu) are not part of the vector: zero, so that the next operand's pad entries are
                R[GQ / 4][GQ % 4] = padl ? 0.0 : R[GQ / 4][GQ % 4];
                store_nu(k);
            });
        (void)NT_;
        wave_sync();
    }

    // ---- the interior-point iteration of qp_solve (below) with the BOUND ROWS IN REGISTERS (round 4; at most 128 rows: two per lane — the chain
    // problems bound the controls only, 3 x 40 rows).  Every row phase of qp_solve below is a pass over the rows through the
    // workspace: multipliers, slacks, the row's entry of the iterate and of the direction — a global-memory round trip (~2 us with
    // the chip streaming) per phase, ~27 of them per iteration, one lane-pass each.  Here a lane keeps its rows' (lam, t, aff, value,
    // residual entry) for the whole QP; what the sweeps need (barrier diagonal, modified gradient at the rows) is stored, the
    // direction at the rows is the one load per sweep, and the dense vector updates of an iteration are one fused pass.
    template <class HS>
    MPCRL_DI bool qp_solve_rows(HS &hs, const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu, bool rg_ready) {
        const bool warm = warm_mu > 0.0;
        const int ne = (N + 1) * NW;
        constexpr bool MERGED = MPCRL_CHAIN_MERGE_CALLS != 0 && USE_V2 && std::is_same<HS, HessConst<M>>::value;
        for (int e = lane; e < (N + 1) * NX; e += NT) dx[e] = e < NX ? x0[e] - X[e] : 0.0, nuq[e] = warm ? NUv[e] : 0.0;
        for (int e = lane; e < N * NU; e += NT) du[e] = (qmode && e < NU) ? u0f[e] - U[e] : 0.0;
        wave_sync();
        // ---- this lane's rows
        bool on[2], hs_[2][2];
        int re_[2];
        unsigned doff[2], Doff[2];      // where the row's entry of (dx | du) and (Dx | Du) sits, relative to dx / Dx
        double lb_[2], ub_[2], v0[2], dvq[2], rgr[2], lm[2][2], tt_[2][2], af[2][2];
        double cnt = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r_ = lane + 64 * j;
            on[j] = r_ < nrows;
            int k = 0, i = 0;
            if (on[j]) row_of(r_, k, i);
            re_[j] = k * NW + i;
            lb_[j] = lbv(k, i), ub_[j] = ubv(k, i);
            hs_[j][0] = on[j] && has(0, k, i), hs_[j][1] = on[j] && has(1, k, i);
            const bool isu = i < NU;
            doff[j] = isu ? (du.off - dx.off) + (unsigned)((k < N ? k : 0) * NU + i) : (unsigned)(k * NX + i - NU);
            Doff[j] = isu ? (Du.off - Dx.off) + (unsigned)((k < N ? k : 0) * NU + i) : (unsigned)(k * NX + i - NU);
            v0[j] = on[j] ? vc(k, i) : 0.0;
            dvq[j] = on[j] ? dx[(int)doff[j]] : 0.0;
            rgr[j] = on[j] ? rg[re_[j]] : 0.0;
            const double v = v0[j] + dvq[j];
            double drg = 0.0;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                lm[j][sd] = 0.0, tt_[j][sd] = 1.0, af[j][sd] = 0.0;
                if (hs_[j][sd]) {
                    cnt += 1.0;
                    const double l_old = LAM(sd, re_[j]), sl = sd ? ub_[j] - v : v - lb_[j];
                    double l, t1;
                    if (warm) {
                        l = l_old, t1 = fmax(sl, TT(sd, re_[j]));
                        if (l * t1 < warm_mu) {
                            if (l >= t1)
                                t1 = warm_mu / l;
                            else
                                l = warm_mu / t1;
                        }
                    } else {
                        t1 = fmax(sl, IPM_T_MIN);
                        l = IPM_MU0 / t1;
                    }
                    lm[j][sd] = l, tt_[j][sd] = t1;
                    drg += sd ? l - l_old : l_old - l;
                }
            }
            if (rg_ready && on[j] && !skipc(k, i) && !fixedc(k, i)) {
                rgr[j] += drg;
                rg[re_[j]] = rgr[j];
            }
        }
        auto store_rows = [&]() {      // multipliers and slacks back to the workspace (next QP's warm start, the kernel's write-out)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) LAM(sd, re_[j]) = lm[j][sd], TT(sd, re_[j]) = tt_[j][sd];
        };
        const double n_rows = wave_sum(cnt);
        if (!rg_ready) store_rows();      // (qp_residuals reads lam from the workspace)
        wave_sync();
        bool ok = false, stepped = false;
        double rlin = wave_max(rg_ready ? qp_start_residuals() : qp_residuals_call(ctx()));
        if (!rg_ready) {
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] = on[j] ? rg[re_[j]] : 0.0;
        } else {      // the stage-0 terms of a warm solve from a new state may have touched this lane's entries
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] = (on[j] && re_[j] < NW) ? rg[re_[j]] : rgr[j];
        }
        // rt = rg, Dg = 0 once: the rows rewrite their own entries in every pass, the other entries of rt follow rg in the fused update
        batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int e, double v) { rt[e] = v, Dg[e] = 0.0; });
        wave_sync();
        for (int it = 0;; ++it) {
            ph(7);
            double rloc = rlin, muloc = 0.0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double v = v0[j] + dvq[j];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) {
                        rloc = fmax(rloc, fabs(tt_[j][sd] - (sd ? ub_[j] - v : v - lb_[j])));
                        muloc = fma(lm[j][sd], tt_[j][sd], muloc);
                    }
            }
            const double rinf = wave_max(rloc);
            const double mu = n_rows > 0.0 ? wave_sum(muloc) / n_rows : 0.0;
            if (rinf <= tol_res && mu <= tol_mu) {
                ok = true;
                break;
            }
            if (it >= IPM_MAX_ITER || !(rinf < 1e300)) break;
            ++n_it;
            ph(0);
            double sigma_mu = 0.0, alpha = 1.0, dvr[2] = {0.0, 0.0};
            bool fail = false;
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double dg = 0.0, er = 0.0;
                    const double v = v0[j] + dvq[j];
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (hs_[j][sd]) {
                            const double l1 = lm[j][sd], t1 = tt_[j][sd];
                            const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                            const double rm = fma(l1, t1, pass ? af[j][sd] - sigma_mu : 0.0);
                            dg += l1 / t1;
                            er += (sd ? -1.0 : 1.0) * (rm - l1 * rd1) / t1;
                        }
                    if (on[j]) {
                        if (pass == 0) Dg[re_[j]] = dg;
                        rt[re_[j]] = rgr[j] + er;
                    }
                }
                wave_sync();
                ph(1);
                if constexpr (MERGED) {
                    if (pass == 0) {
                        if (!pred_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                    } else
                        corr_call(ctx(), rt.off, rb.off);
                } else {
                    if (pass == 0) {
                        if (!factor_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                        ph(2);
                    } else {
                        backward_vec_call(ctx(), rt.off);
                        ph(3);
                    }
                    forward_call<false>(ctx(), rb.off);
                }
                ph(4);
#pragma unroll
                for (int j = 0; j < 2; ++j) dvr[j] = on[j] ? Dx[(int)Doff[j]] : 0.0;      // the direction at the rows: the one load of the pass
                double amax = 1.0, muaff = 0.0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd)
                        if (hs_[j][sd]) {
                            const double l1 = lm[j][sd], t1 = tt_[j][sd];
                            const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                            const double rm = fma(l1, t1, pass ? af[j][sd] - sigma_mu : 0.0);
                            const double dt1 = -rd1 + (sd ? -dv : dv);
                            const double dl1 = (-rm - l1 * dt1) / t1;
                            if (dl1 < 0.0) amax = fmin(amax, -l1 / dl1);
                            if (dt1 < 0.0) amax = fmin(amax, -t1 / dt1);
                        }
                }
                amax = -wave_max(-amax);
                if (pass == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd)
                            if (hs_[j][sd]) {
                                const double l1 = lm[j][sd], t1 = tt_[j][sd];
                                const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                                const double dt1 = -rd1 + (sd ? -dv : dv);
                                const double dl1 = (-l1 * t1 - l1 * dt1) / t1;
                                muaff = fma(fma(amax, dl1, l1), fma(amax, dt1, t1), muaff);
                                af[j][sd] = dl1 * dt1;
                            }
                    }
                    const double mu_aff = n_rows > 0.0 ? wave_sum(muaff) / n_rows : 0.0;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                } else
                    alpha = fmin(1.0, fmax(IPM_FRAC, 1.0 - mu) * amax);   // fraction to the boundary -> 1 as mu -> 0
            }
            ph(5);
            if (fail) break;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double v = v0[j] + dvq[j], dv = dvr[j];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (hs_[j][sd]) {
                        const double l1 = lm[j][sd], t1 = tt_[j][sd];
                        const double rd1 = t1 - (sd ? ub_[j] - v : v - lb_[j]);
                        const double rm = fma(l1, t1, af[j][sd] - sigma_mu);
                        const double dt1 = -rd1 + (sd ? -dv : dv);
                        const double dl1 = (-rm - l1 * dt1) / t1;
                        lm[j][sd] = fma(alpha, dl1, l1);
                        tt_[j][sd] = fma(alpha, dt1, t1);
                    }
                dvq[j] = fma(alpha, dvr[j], dvq[j]);      // (the same fma the dense update below applies to the entry)
            }
            // one fused pass: dx += alpha Dx, du += alpha Du, rg *= (1 - alpha) (and rt = rg), rb *= (1 - alpha)
            const double om = 1.0 - alpha;
            struct Upd {
                double a, b, c, d, e, f;
            };
            batched_pass<4>(ne, lane,
                            [&](int e) {
                                const int ex = e < (N + 1) * NX ? e : 0, eu = e < N * NU ? e : 0, eb = e < N * NX ? e : 0;
                                return Upd{rg[e], Dx[ex], dx[ex], Du[eu], du[eu], rb[eb]};
                            },
                            [&](int e, const Upd &v) {
                                const double g = om * v.a;
                                rg[e] = g, rt[e] = g;
                                if (e < (N + 1) * NX) dx[e] = fma(alpha, v.b, v.c);
                                if (e < N * NU) du[e] = fma(alpha, v.d, v.e);
                                if (e < N * NX) rb[e] = om * v.f;
                            });
#pragma unroll
            for (int j = 0; j < 2; ++j) rgr[j] *= om;
            rlin *= om;
            stepped = true;
            wave_sync();
        }
        store_rows();
        wave_sync();
        if (stepped) costate_call(ctx());      // nuq of the point the iteration ended at (no iteration: the warm multipliers stand)
        return ok;
    }

    // ---- Mehrotra predictor-corrector on the QP of the current linearisation (hard bounds) --------------------
    // rg_ready: rg holds q -+ lam + [B A]' NUv - [0; NUv] of this linearisation (round_start left it there) and the QP starts from
    // nuq = NUv (warm) or from NUv = 0 (cold): its starting residual needs no pass over the [B A]_k
    template <class HS>
    MPCRL_DI bool qp_solve(HS &hs, const double *x0, const double *u0f, int &n_it, double warm_mu, double tol_res, double tol_mu, bool rg_ready = false) {
        if constexpr (USE_V2 && MPCRL_CHAIN_SCALE_RES && MPCRL_CHAIN_ROWS_IN_REGS)
            if (nrows <= 128) return qp_solve_rows(hs, x0, u0f, n_it, warm_mu, tol_res, tol_mu, rg_ready);
        const bool warm = warm_mu > 0.0;
        const int ne = (N + 1) * NW;
        for (int e = lane; e < (N + 1) * NX; e += NT) dx[e] = e < NX ? x0[e] - X[e] : 0.0, nuq[e] = warm ? NUv[e] : 0.0;
        for (int e = lane; e < N * NU; e += NT) du[e] = (qmode && e < NU) ? u0f[e] - U[e] : 0.0;
        wave_sync();
        double cnt = 0.0;
        for (int r_ = lane; r_ < nrows; r_ += NT) {
            int k, i;
            row_of(r_, k, i);
            const int e = k * NW + i;
            const double v = vc(k, i) + dvc(dx, du, k, i);
            double drg = 0.0;      // the multipliers of this row change: so does its entry of the stationarity residual
            for (int sd = 0; sd < 2; ++sd)
                if (has(sd, k, i)) {
                    cnt += 1.0;
                    const double l_old = LAM(sd, e);
                    double l_new;
                    if (warm) {
                        double l = l_old, tt = fmax(bslack(sd, k, i, v), TT(sd, e));
                        if (l * tt < warm_mu) {
                            if (l >= tt)
                                tt = warm_mu / l;
                            else
                                l = warm_mu / tt;
                        }
                        LAM(sd, e) = l, TT(sd, e) = tt;
                        l_new = l;
                    } else {
                        const double tt = fmax(bslack(sd, k, i, v), IPM_T_MIN);
                        TT(sd, e) = tt;
                        l_new = IPM_MU0 / tt;
                        LAM(sd, e) = l_new;
                    }
                    drg += sd ? l_new - l_old : l_old - l_new;
                }
            if (rg_ready && !skipc(k, i) && !fixedc(k, i)) rg[e] += drg;
        }
        const double n_rows = wave_sum(cnt);
        wave_sync();
        bool ok = false, stepped = false;
        double rlin = 0.0;
        for (int it = 0;; ++it) {
            ph(7);
            // The residuals of the LINEAR equations (dynamics rb, stationarity rg) are evaluated once per QP: a step of length alpha
            // along a direction that solves the Newton system takes them to (1 - alpha) times their value, exactly — they are
            // scaled at the end of the iteration instead of being re-evaluated (a sweep over all [B A]_k: 161 KB per instance at
            // n_mass 5, 10 % of the kernel).  Only the bound rows below depend on the step nonlinearly (complementarity).
            if (!MPCRL_CHAIN_SCALE_RES || it == 0) rlin = wave_max((rg_ready && it == 0) ? qp_start_residuals() : qp_residuals_call(ctx()));
            double rloc = rlin, muloc = 0.0;
            for (int r_ = lane; r_ < nrows; r_ += NT) {
                int k, i;
                row_of(r_, k, i);
                const int e = k * NW + i;
                const double v = vc(k, i) + dvc(dx, du, k, i);
                for (int sd = 0; sd < 2; ++sd)
                    if (has(sd, k, i)) {
                        rloc = fmax(rloc, fabs(TT(sd, e) - bslack(sd, k, i, v)));
                        muloc = fma(LAM(sd, e), TT(sd, e), muloc);
                    }
            }
            const double rinf = wave_max(rloc);
            const double mu = n_rows > 0.0 ? wave_sum(muloc) / n_rows : 0.0;
            if (rinf <= tol_res && mu <= tol_mu) {
                ok = true;
                break;
            }
            if (it >= IPM_MAX_ITER || !(rinf < 1e300)) break;
            ++n_it;
            ph(0);
            double sigma_mu = 0.0, alpha = 1.0;
            bool fail = false;
            for (int pass = 0; pass < 2; ++pass) {
                // barrier diagonal + modified gradient: rt = rg everywhere, corrected on the bounded rows
                batched_pass<8>(ne, lane, [&](int e) { return rg[e]; },
                                [&](int e, double v) {
                                    rt[e] = v;
                                    if (pass == 0) Dg[e] = 0.0;
                                });
                wave_sync();
                for (int r_ = lane; r_ < nrows; r_ += NT) {
                    int k, i;
                    row_of(r_, k, i);
                    const int e = k * NW + i;
                    double dg = 0.0, er = 0.0;
                    const double v = vc(k, i) + dvc(dx, du, k, i);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has(sd, k, i)) {
                            const double l1 = LAM(sd, e), t1 = TT(sd, e);
                            const double rd1 = t1 - bslack(sd, k, i, v);
                            const double rm = fma(l1, t1, pass ? AFF(sd, e) - sigma_mu : 0.0);
                            dg += l1 / t1;
                            er += (sd ? -1.0 : 1.0) * (rm - l1 * rd1) / t1;
                        }
                    if (pass == 0) Dg[e] = dg;
                    rt[e] = rg[e] + er;
                }
                wave_sync();
                ph(1);
                if (pass == 0) {
                    if (!factor_call<HS>(ctx(), hs.hex_offset(), rt.off, rb.off)) fail = true;
                    ph(2);
                } else {
                    backward_vec_call(ctx(), rt.off);
                    ph(3);
                }
                if (pass == 1 && !USE_V2)   // the multiplier step is only needed with the final direction (round 4: not at all, costate_nu)
                    forward_call<true>(ctx(), rb.off);
                else
                    forward_call<false>(ctx(), rb.off);
                ph(4);
                double amax = 1.0, muaff = 0.0;
                for (int r_ = lane; r_ < nrows; r_ += NT) {
                    int k, i;
                    row_of(r_, k, i);
                    const int e = k * NW + i;
                    const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                    for (int sd = 0; sd < 2; ++sd)
                        if (has(sd, k, i)) {
                            const double l1 = LAM(sd, e), t1 = TT(sd, e);
                            const double rd1 = t1 - bslack(sd, k, i, v);
                            const double rm = fma(l1, t1, pass ? AFF(sd, e) - sigma_mu : 0.0);
                            const double dt1 = -rd1 + (sd ? -dv : dv);
                            const double dl1 = (-rm - l1 * dt1) / t1;
                            if (dl1 < 0.0) amax = fmin(amax, -l1 / dl1);
                            if (dt1 < 0.0) amax = fmin(amax, -t1 / dt1);
                        }
                }
                amax = -wave_max(-amax);
                if (pass == 0) {
                    for (int r_ = lane; r_ < nrows; r_ += NT) {
                        int k, i;
                        row_of(r_, k, i);
                        const int e = k * NW + i;
                        const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                        for (int sd = 0; sd < 2; ++sd)
                            if (has(sd, k, i)) {
                                const double l1 = LAM(sd, e), t1 = TT(sd, e);
                                const double rd1 = t1 - bslack(sd, k, i, v);
                                const double dt1 = -rd1 + (sd ? -dv : dv);
                                const double dl1 = (-l1 * t1 - l1 * dt1) / t1;
                                muaff = fma(fma(amax, dl1, l1), fma(amax, dt1, t1), muaff);
                                AFF(sd, e) = dl1 * dt1;
                            }
                    }
                    const double mu_aff = n_rows > 0.0 ? wave_sum(muaff) / n_rows : 0.0;
                    const double ratio = mu > 0.0 ? mu_aff / mu : 0.0;
                    sigma_mu = ratio * ratio * ratio * mu;
                    wave_sync();
                } else
                    alpha = fmin(1.0, fmax(IPM_FRAC, 1.0 - mu) * amax);   // fraction to the boundary -> 1 as mu -> 0
            }
            ph(5);
            if (fail) break;
            for (int r_ = lane; r_ < nrows; r_ += NT) {
                int k, i;
                row_of(r_, k, i);
                const int e = k * NW + i;
                const double v = vc(k, i) + dvc(dx, du, k, i), dv = dvc(Dx, Du, k, i);
                for (int sd = 0; sd < 2; ++sd)
                    if (has(sd, k, i)) {
                        const double l1 = LAM(sd, e), t1 = TT(sd, e);
                        const double rd1 = t1 - bslack(sd, k, i, v);
                        const double rm = fma(l1, t1, AFF(sd, e) - sigma_mu);
                        const double dt1 = -rd1 + (sd ? -dv : dv);
                        const double dl1 = (-rm - l1 * dt1) / t1;
                        LAM(sd, e) = fma(alpha, dl1, l1);
                        TT(sd, e) = fma(alpha, dt1, t1);
                    }
            }
            wave_sync();
            if constexpr (USE_V2)
                batched_pass<8>((N + 1) * NX, lane, [&](int e) { return Pair2{Dx[e], dx[e]}; },
                                [&](int e, const Pair2 &v) { dx[e] = fma(alpha, v.a, v.b); });
            else
                batched_pass<4>((N + 1) * NX, lane, [&](int e) { return Quad4{Dx[e], dx[e], Dnu[e], nuq[e]}; },
                                [&](int e, const Quad4 &v) { dx[e] = fma(alpha, v.a, v.b), nuq[e] = fma(alpha, v.c, v.d); });
            stepped = true;
            batched_pass<2>(N * NU, lane, [&](int e) { return Pair2{Du[e], du[e]}; },
                            [&](int e, const Pair2 &v) { du[e] = fma(alpha, v.a, v.b); });
            if (MPCRL_CHAIN_SCALE_RES) {
                const double om = 1.0 - alpha;
                batched_pass<8>(ne, lane, [&](int e) { return rg[e]; }, [&](int e, double v) { rg[e] = om * v; });
                batched_pass<8>(N * NX, lane, [&](int e) { return rb[e]; }, [&](int e, double v) { rb[e] = om * v; });
                rlin *= om;
            }
            wave_sync();
        }
        if constexpr (USE_V2)
            if (stepped) costate_call(ctx());      // nuq of the point the iteration ended at (no iteration: the warm multipliers stand)
        return ok;
    }

    // ---- phase calls.  The big phases are real (non-inlined) functions: each gets a register allocation of its own, so the
    // operands of one phase are never spilled on behalf of another (inlined into one body, the interior-point loop carried
    // hundreds of hoisted loop invariants through every phase, and each reload from scratch is an s_waitcnt vmcnt(0) that also
    // drains the streaming stores).  What travels is a CONTEXT of 16 dwords (argument registers): workspace, parameters, iterate,
    // LDS addresses, horizon.  Until round 3 the solver itself travelled by value — ~140 dwords per lane, i.e. 35 KB per wavefront
    // written to and read back from scratch memory at every call (4.7 KB per lane of frame, a large part of the kernel's HBM
    // traffic beyond its streamed factors); every field of it is a function of the context: the workspace arrays are offsets of one
    // base (LargeLayout), the tile indices functions of the lane, the row counts three words the set-up left in LDS.
    MPCRL_DI static unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
    template <class T>
    MPCRL_DI static T *uni_global(T *ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = rfl((unsigned)v), hi = rfl((unsigned)(v >> 32));
        typedef __attribute__((address_space(1))) T GT;
        return (T *)(GT *)(((unsigned long long)hi << 32) | lo);
    }
    template <class T>
    MPCRL_DI static T *uni_lds(T *ptr) {
        typedef __attribute__((address_space(3))) T LT;
        return (T *)(LT *)(unsigned long)rfl((unsigned)(unsigned long long)ptr);
    }
    MPCRL_DI void uni(WsArr &a_) const { a_.base = a_.base ? BA.base : nullptr, a_.off = rfl(a_.off); }
    struct Ctx {
        double *w, *X, *U;
        const double *th, *xs;
        double *lds;
        int *sidx;
        int N, qmode;
#ifdef MPCRL_PROFILE_PHASES
        unsigned long long *ph_lds;
#endif
    };
    MPCRL_DI Ctx ctx() const {
        Ctx c;
        c.w = (double *)BA.base, c.X = X, c.U = U, c.th = th, c.xs = xs, c.lds = lds, c.sidx = sidx, c.N = N, c.qmode = qmode ? 1 : 0;
#ifdef MPCRL_PROFILE_PHASES
        c.ph_lds = ph_lds;
#endif
        return c;
    }
    // the solver of a phase, rebuilt from the context; wave-uniform values end up in scalar registers (readfirstlane on entry,
    // everything derived from them is scalar arithmetic), pointers get their address spaces back
    MPCRL_DI static ChainSolver from_ctx(const Ctx &c) {
        ChainSolver S(uni_global(c.xs), (int)rfl((unsigned)c.N), (int)threadIdx.x);
        S.qmode = rfl((unsigned)c.qmode) != 0;
        S.th = uni_global(c.th), S.X = uni_global(c.X), S.U = uni_global(c.U);
        S.bind_workspace(uni_global(c.w), LargeLayout<M>(S.N));
        S.setup_lane(uni_lds(c.lds), uni_lds(c.sidx), true);
#ifdef MPCRL_PROFILE_PHASES
        S.ph_lds = uni_lds(c.ph_lds);
        S.ph_t = clock64();
#endif
        return S;
    }
    MPCRL_DI WsArr arr(unsigned off) const { return WsArr{BA.base, rfl(off)}; }
    struct RoundStart {
        double cost, res[4];
    };
    __device__ MPCRL_PHASE_FN static RoundStart round_start_call(Ctx c, const double *x0, const double *u0f) {
        ChainSolver S = from_ctx(c);
        x0 = uni_global(x0), u0f = u0f ? uni_global(u0f) : nullptr;
        RoundStart o;
        o.cost = S.round_start(x0, u0f, o.res);
        return o;
    }
    // the whole start of an SQP round as ONE call: parameter / multiplier staging, point pass, direction pass, round_start.  As
    // three calls their prologues and epilogues moved ~70 KB per wavefront and round through scratch (MPCRL_CHAIN_MERGE_CALLS).
    __device__ MPCRL_PHASE_FN static RoundStart round_call(Ctx c, const double *x0, const double *u0f, double h, int steps) {
        ChainSolver S = from_ctx(c);
        using DC_ = DirCfg<M>;
        double *const big = S.lds + Cfg::oBig;
        const int N_ = S.N, lane_ = S.lane;
        for (int e = lane_; e < M::NTD; e += NT) big[DC_::CO + e] = S.th[M::td_index(e)];
        if constexpr (Cfg::FUSE_GT)
            batched_pass<8>((N_ + 1) * NX, lane_, [&](int e) { return S.NUv[e]; }, [&](int e, double v) { big[DC_::CO + M::NTD + e] = v; });
        wave_sync();
        if (lane_ < N_)
            chain_point_body<M, false, true>(S.X, S.U, big + DC_::CO, (double *)S.BA.base, N_, lane_, h, steps,
                                             big + DC_::CO + M::NTD + (Cfg::FUSE_GT ? (N_ + 1) * NX : 0));
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        chain_dir_body<M>(S.th, (double *)S.BA.base, big, N_, lane_, h, steps);
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        RoundStart o;
        o.cost = S.round_start(x0, u0f, o.res);
        return o;
    }
    __device__ MPCRL_PHASE_FN static double qp_residuals_call(Ctx c) {
        ChainSolver S = from_ctx(c);
        if constexpr (USE_V2 && MPCRL_CHAIN_SCALE_RES)
            return S.qp_residuals2();
        else
            return S.qp_residuals();
    }
    // hex_off: workspace offset of the exact Hessian blocks (HessGlobal); the constant Hessian (HessConst) is rebuilt from theta
    template <class HS>
    __device__ MPCRL_PHASE_FN static bool factor_call(Ctx c, unsigned hex_off, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        if constexpr (std::is_same<HS, HessConst<M>>::value ? USE_V2 : USE_V2_SENS) {
            typename HessV2<HS>::type hs;
            hs.init(S, hex_off);
            return S.template factor2<typename HessV2<HS>::type, !std::is_same<HS, HessConst<M>>::value>(hs, S.arr(g_off), S.arr(bb_off));
        } else {
            HS hs;
            hs.init(S, hex_off);
            return S.factor(hs, S.arr(g_off), S.arr(bb_off));
        }
    }
    __device__ MPCRL_PHASE_FN static void backward_vec_call(Ctx c, unsigned g_off) {
        ChainSolver S = from_ctx(c);
        if constexpr (USE_V2)
            S.backward_vec2(S.arr(g_off));
        else
            S.backward_vec(S.arr(g_off));
    }
    __device__ MPCRL_PHASE_FN static void forward_sens_call(Ctx c, unsigned ydx, unsigned ydu, unsigned ydnu) {
        ChainSolver S = from_ctx(c);
        S.forward2_sens(S.arr(ydx), S.arr(ydu), S.arr(ydnu));
    }
    __device__ MPCRL_PHASE_FN static void costate_call(Ctx c) {
        ChainSolver S = from_ctx(c);
        S.costate_nu();
    }
    // predictor and corrector as one call each (round-4 sweeps of the SQP only).  Every phase call saves and restores the callee-saved
    // half of the registers its body uses — 29 KB per wavefront for the factor sweep, ~20 KB for a vector sweep, through scratch, i.e.
    // HBM traffic at 1024 resident wavefronts: ~1.9 GB written and read back per step of 1024 solves at n_mass 5 with four calls per
    // interior-point iteration.  The time is the same either way (measured: 8.68 vs 8.69 ms), the bytes are not.
    template <class HS>
    __device__ MPCRL_PHASE_FN static bool pred_call(Ctx c, unsigned hex_off, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        typename HessV2<HS>::type hs;
        hs.init(S, hex_off);
        const bool ok = S.template factor2<typename HessV2<HS>::type, false>(hs, S.arr(g_off), S.arr(bb_off));
        S.template forward2<false>(S.arr(bb_off));
        return ok;
    }
    __device__ MPCRL_PHASE_FN static void corr_call(Ctx c, unsigned g_off, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        S.backward_vec2(S.arr(g_off));
        S.template forward2<false>(S.arr(bb_off));
    }
    template <bool want_nu>
    __device__ MPCRL_PHASE_FN static void forward_call(Ctx c, unsigned bb_off) {
        ChainSolver S = from_ctx(c);
        if constexpr (want_nu ? USE_V2_SENS : USE_V2)      // (with the multiplier step: only the sensitivities' adjoint solves)
            S.template forward2<want_nu>(S.arr(bb_off));
        else
            S.template forward<want_nu>(S.arr(bb_off));
    }

    MPCRL_DI void bind_workspace(double *w, const LargeLayout<M> &lay) {
        auto at = [&](size_t o) { return WsArr{(char *)w, (unsigned)o}; };
        BA = at(lay.BA), r = at(lay.r), q = at(lay.q), dx = at(lay.dx), du = at(lay.du), nuq = at(lay.nuq);
        Dx = at(lay.Dx), Du = at(lay.Du), Dnu = at(lay.Dnu), rg = at(lay.rg), rb = at(lay.rb), rt = at(lay.rt), Dg = at(lay.Dg);
        lam = at(lay.lamw), t = at(lay.tw), aff = at(lay.aff), P = at(lay.P), p = at(lay.p), K = at(lay.K), L = at(lay.L);
        kff = at(lay.kff), Acl = at(lay.Acl), hb = at(lay.hb), ccv = at(lay.ccv), cvec = at(lay.cvec), NUv = at(lay.ynu), state = at(lay.state);
        G2 = at(lay.G2), P2 = at(lay.P2), hb2 = at(lay.hb2), minv2 = at(lay.minv2), mvu2 = at(lay.mvu2);
    }
};

// =====================================================================================================
// SQP, as a sequence of launches:  init, then (lin, qp) x (max_iter + 1).  Per-instance state lives in ws.state.
// =====================================================================================================

// ---- iterate set-up: cold start (MPC.reset, mpc.py:204-210) or the stored one.  One wavefront per instance.
template <class M>
__global__ void __launch_bounds__(64) chain_init_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NT = 64;
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU;
    const double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    const double *bnd = a.BND + (size_t)inst * 10 * nb;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = a.u0fix ? a.u0fix + (size_t)inst * NU : nullptr;
    double *NUv = w + lay.ynu, *lam = w + lay.lamw, *t = w + lay.tw, *aff = w + lay.aff, *st = w + lay.state;
    const int ne = (N + 1) * NW;
    double stepn = -1.0;   // perturbation seen by the first QP (< 0: cold)
    if ((a.flags & 8) || (a.cold && a.cold[inst])) {
        for (int e = lane; e < (N + 1) * NX; e += NT) X[e] = x0[e % NX], NUv[e] = 0.0;
        for (int e = lane; e < N * NU; e += NT) U[e] = 0.0;
        for (int e = lane; e < 2 * ne; e += NT) lam[e] = 0.0, t[e] = 1.0, aff[e] = 0.0;
    } else {
        for (int e = lane; e < (N + 1) * NX; e += NT) NUv[e] = e < NX ? 0.0 : PIg[e - NX];
        for (int e = lane; e < 2 * ne; e += NT) lam[e] = bnd[e], t[e] = bnd[2 * nb + e], aff[e] = 0.0;
        double sl = 0.0;
        if (lane < NX) sl = fabs(x0[lane] - X[lane]);
        if (u0f && lane < NU) sl = fmax(sl, fabs(u0f[lane] - U[lane]));
        stepn = (a.flags & 16) ? -1.0 : wave_max(sl);   // MPCRL_COLD_DUAL: the interior point starts from its default point
    }
    if (lane == 0) {
        st[ST_ACTIVE] = 1.0, st[ST_IT] = 0.0, st[ST_NIPM] = 0.0, st[ST_TIGHT] = 1.0, st[ST_STEPN] = stepn, st[ST_COST] = 0.0;
        st[ST_STATUS] = 2.0;
        for (int j = 0; j < 4; ++j) st[ST_RES + j] = 0.0;
    }
}

// ---- the POINT pass of the derivative kernels: one lane per (instance, stage) walks the 4 x rk_steps evaluation points of the RK4
// map in plain doubles and leaves, per evaluation point and link, what the tangent of the ODE needs there (ChainDev::ode_coef) in the
// instance's workspace; it also writes r_k = F(x_k, u_k) - x_{k+1}.  With SECOND (sensitivities) the tables carry the second-order
// coefficients as well and a reverse sweep of nu_{k+1} through the same points adds the 3 x 3 Hessian of every link force
// (ChainDev::link_hessian).  The direction kernels below (one lane per direction) then start from these tables: done inside them, this
// pass was repeated by every lane of a stage — 60 % of the linearisation's and half of the Hessian kernel's instructions.
// the pass itself, for stage k of one instance (X, U, th: the instance's iterate and parameters, w: its workspace).  A real call: the
// QP kernel runs it at the end of a round on N of its lanes, with a register allocation of its own.
// TH_LDS (the SQP kernel): th is the instance's COMPACT parameter copy in LDS (the NTD differentiable entries, staged by the caller) —
// out of the parameter vector in global memory every evaluation point fetched its ~50-75 coefficients again, behind the table
// stores of the point before.
template <class M, bool SECOND, bool TH_LDS>
MPCRL_DI void chain_point_body(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_);
template <class M, bool SECOND, bool TH_LDS>
__device__ MPCRL_PHASE_FN void chain_point_pass(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_) {
    chain_point_body<M, SECOND, TH_LDS>(X_, U_, th_, w_, N, k, h, steps, lacc_);
}
template <class M, bool SECOND, bool TH_LDS>
MPCRL_DI void chain_point_body(const double *X_, const double *U_, const double *th_, double *w_, int N, int k, double h, int steps, double *lacc_) {
    constexpr int NX = M::NX, NU = M::NU, NL = M::NL, TS = SECOND ? M::TAB2 : M::TAB;
    const double *X = as_global(X_), *U = as_global(U_), *th = TH_LDS ? as_lds(th_) : as_global(th_);
    double *w = as_global(w_);
    const LargeLayout<M> lay(N);
    double *tab = w + lay.ptab + (size_t)k * 8 * NL * M::TAB2;
    double u[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = U[k * NU + i];
    struct AccReg {
        double a[NX];
        MPCRL_DI double &operator[](int i) { return a[i]; }
    };
    struct AccLds {
        double *p;
        int st;
        MPCRL_DI double &operator[](int i) const { return p[i * st]; }
    };
    {
        // (TH_LDS: the RK4 accumulator of the lane lives in LDS, entry i at lacc[i N + k].  With all four arrays in registers the
        // compiler kept ~8 doubles of them in scratch, and every reload — an s_waitcnt vmcnt(0) — also waited for the table stores in
        // flight, 40 scattered lines each: ~20 drains per RK4 step were most of this pass's time.)
        std::conditional_t<TH_LDS, AccLds, AccReg> acc;
        if constexpr (TH_LDS) acc.p = as_lds(lacc_) + k, acc.st = N;
        double xc[NX], kk[NX], xt[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = X[k * NX + i];
        // (with SECOND also the velocity difference of every link at every evaluation point: the mixed term wants it, chain_sens_mix2)
        auto store_dv = [&](const double *xs_, int e) {
            if constexpr (SECOND) {
                double *qv = w + lay.qvtab + ((size_t)k * 8 + e) * NL * 6;
                constexpr int Mm = M::M;
#pragma unroll
                for (int i = 0; i < NL; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double vr = i < Mm ? xs_[3 * (Mm + 1) + 3 * (i < Mm ? i : 0) + j] : u[j];
                        qv[6 * i + 3 + j] = i ? vr - xs_[3 * (Mm + 1) + 3 * (i > 0 ? i - 1 : 0) + j] : vr;
                    }
            }
        };
        for (int s = 0; s < steps; ++s) {
            double *tb = tab + (size_t)(4 * s) * NL * TS;
            M::template ode_coef<SECOND, TH_LDS>(xc, u, th, kk, tb, true);
            store_dv(xc, 4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + NL * TS, true);
            store_dv(xt, 4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + (0.5 * h) * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + 2 * NL * TS, true);
            store_dv(xt, 4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * kk[i], xt[i] = xc[i] + h * kk[i];
            M::template ode_coef<SECOND, TH_LDS>(xt, u, th, kk, tb + 3 * NL * TS, true);
            store_dv(xt, 4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xc[i] + (h / 6.0) * (acc[i] + kk[i]);
        }
        if constexpr (!SECOND) {
            double *r = w + lay.r + k * NX;
#pragma unroll
            for (int i = 0; i < NX; ++i) r[i] = xc[i] - X[(k + 1) * NX + i];
        }
    }
    if constexpr (SECOND) {   // the adjoint: kb_e = d(nu' F) / d(k_e) at every evaluation point, last step first; G_{e,i} from its force part
        const double *nu = w + lay.ynu;
        double *Gt = w + lay.gtab + (size_t)k * 8 * NL * 6;
        std::conditional_t<TH_LDS, AccLds, AccReg> acc;
        if constexpr (TH_LDS) acc.p = as_lds(lacc_) + k, acc.st = N;
        double lb[NX], kb[NX], Xb[NX], q[3 * NL];
#pragma unroll
        for (int i = 0; i < NX; ++i) lb[i] = nu[(k + 1) * NX + i];
        for (int s = steps - 1; s >= 0; --s) {
            auto node = [&](int e) {   // Xb = J(e)' kb, and the link Hessians of this evaluation point
                const double *tb = tab + (size_t)e * NL * TS;
#pragma unroll
                for (int i = 0; i < NX; ++i) Xb[i] = 0.0;
                M::template ode_tan_T<TS>(tb, th, kb, Xb, q);
                double *qv = w + lay.qvtab + ((size_t)k * 8 + e) * NL * 6;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    M::link_hessian(tb + i * TS, q + 3 * i, Gt + ((size_t)e * NL + i) * 6);
#pragma unroll
                    for (int j = 0; j < 3; ++j) qv[6 * i + j] = q[3 * i + j];
                }
            };
#pragma unroll
            for (int i = 0; i < NX; ++i) kb[i] = (h / 6.0) * lb[i];
            node(4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = lb[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + h * Xb[i];
            node(4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 3.0) * lb[i] + (0.5 * h) * Xb[i];
            node(4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + Xb[i], kb[i] = (h / 6.0) * lb[i] + (0.5 * h) * Xb[i];
            node(4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) lb[i] = acc[i] + Xb[i];
        }
    }
}

// One wavefront per instance, lane = stage (N <= 64): the instance's differentiable parameters and the lanes' RK4 accumulators sit in LDS
// (as in the SQP kernel's call).  Until round 4 the lanes of a wavefront ran over (instance, stage) pairs with everything in
// registers: ~50 doubles of them in scratch, every reload waiting for the scattered table stores in flight — 261 us at n_mass 5 for a
// pass that takes 35 us inside the SQP kernel.
template <class M, bool SECOND>
__global__ void __launch_bounds__(64) chain_point_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU;
    extern __shared__ __attribute__((aligned(16))) double lds[];   // NTD + N NX doubles
    const int N = sp.N, inst = blockIdx.x, lane = threadIdx.x;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    if constexpr (SECOND) {
        const int status = a.status[inst];
        if (!(status == 0 || status == 2)) return;
    } else {
        if (w[lay.state + ST_ACTIVE] == 0.0) return;
    }
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    for (int e = lane; e < M::NTD; e += 64) lds[e] = th[M::td_index(e)];
    wave_sync();
    if (lane < N)
        chain_point_pass<M, SECOND, true>(a.X + (size_t)inst * (N + 1) * NX, a.U + (size_t)inst * N * NU, lds, w, N, lane, sp.h, sp.rk_steps, lds + M::NTD);
}

// ---- dynamics linearisation, the DIRECTION pass: fills [B A]_k of all stages of ONE instance, run by the instance's own wavefront
// inside the SQP kernel.  One lane per (stage, direction), 64 of them per step; the coefficient tables (chain_point_pass) of the
// stages a step touches are copied to LDS, then each lane propagates ONLY its tangent through the evaluation points.  A forward jet
// per lane (value + tangent through the whole map) needs both sets of arrays live at once — 2 x 4 NX doubles, 528 registers at
// NX = 33, i.e. spills whose scratch traffic made the round-1 kernel HBM-bound (46 GB per step at n_mass = 7) — and recomputes the
// point NW times.  As a grid-wide kernel of its own (first half of round 2) it cost the same SIMD time — an instance's 40 x NW
// directions are 15 (23) wavefront-steps either way, and a batch of 1024 is one wavefront per SIMD — plus a kernel boundary per round.
template <class M>
struct DirCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, EV = 8;
    static constexpr int TSZ = EV * NL * M::TAB;                    // doubles of one stage's first-order table in the workspace
    // In LDS a link's record is 12 doubles — its 9 coefficients and the link's 3 damping coefficients — on a 16-byte boundary: six
    // ds_read_b128 per link (4 LDS cycles each) where the 9 + 3 doubles at odd offsets were five ds_read2_b64 (8 cycles each: the
    // instruction runs at half the LDS rate) and two ds_read_b64.  Four wavefronts per CU run this pass at the same time and it is
    // bound by the LDS pipe they share.
    // (+ 8 doubles between the stages' tables: EV NL LREC doubles is a multiple of the 64 banks for every chain size, so the lanes of
    // two stages in one ds_read_b128 lane group read different addresses on the SAME banks — a 2-way conflict in three of the four
    // groups of a step; 16 dwords apart they do not meet)
    static constexpr int LREC = 12, TSZL = EV * NL * LREC + 8;
    static constexpr int SPAN = (64 + NW - 1) / NW + 1;              // stages a step of 64 consecutive (stage, direction) items can touch
    static constexpr bool FITS = SPAN * TSZL <= 2048;                // else whole stages per step
    static constexpr int NST = FITS ? SPAN : 64 / NW;                // stage tables in LDS
    static constexpr int LP = FITS ? 64 : (64 / NW) * NW;            // items per step
    static constexpr int CO = NST * TSZL;                            // after the tables: the instance's NTD differentiable parameters
    static_assert(ChainCfg<M>::oBig % 2 == 0, "16-byte records");
};

template <class M>
MPCRL_DI void chain_dir_body(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps);
template <class M>
__device__ MPCRL_PHASE_FN void chain_dir_pass(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps) {
    chain_dir_body<M>(th_, w_, tabl_, N, lane, h, steps);
}
template <class M>
MPCRL_DI void chain_dir_body(const double *th_, double *w_, double *tabl_, int N, int lane, double h, int steps) {
    using DC = DirCfg<M>;
    const double *th = as_global(th_);
    double *w = as_global(w_), *tabl = as_lds(tabl_);
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB = M::TAB, TSZ = DC::TSZ, STG = DC::EV * NL * M::TAB2;
    const LargeLayout<M> lay(N);
    const int items = N * NW;
    // the tables of step i + 1 are requested before step i is computed and go to LDS after it: one global round trip per step hidden
    constexpr int NPF = (DC::NST * TSZ + 63) / 64, LREC = DC::LREC, TSZL = DC::TSZL;
    double pf[NPF];
    auto request = [&](int i0) {
        const int k_lo = i0 / NW, cnt = (min(N - 1, (i0 + DC::LP - 1) / NW) - k_lo + 1) * TSZ;
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int e = lane + 64 * j, ks = e / TSZ;
            pf[j] = e < cnt ? w[lay.ptab + (size_t)(k_lo + ks) * STG + (e - ks * TSZ)] : 0.0;
        }
    };
    request(0);
    // the damping coefficients (out of the compact parameter copy the caller staged) go into every link record once: a record's
    // link index is a function of its position
    for (int r = lane; r < DC::NST * DC::EV * NL; r += 64) {
        const int st_ = r / (DC::EV * NL), rr = r - st_ * (DC::EV * NL);
#pragma unroll
        for (int j = 0; j < 3; ++j) tabl[st_ * TSZL + rr * LREC + 9 + j] = tabl[DC::CO + 7 * NL + 3 * (rr % NL) + j];
    }
    for (int i0 = 0; i0 < items; i0 += DC::LP) {
        const int k_lo = i0 / NW, k_hi = min(N - 1, (i0 + DC::LP - 1) / NW);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int e = lane + 64 * j;
            const int ks = e / TSZ, idx = e - ks * TSZ, rec = idx / M::TAB;
            if (e < (k_hi - k_lo + 1) * TSZ) tabl[ks * TSZL + rec * LREC + (idx - rec * M::TAB)] = pf[j];
        }
        wave_sync();
        if (i0 + DC::LP < items) request(i0 + DC::LP);
        const int it_ = i0 + lane;
        const bool on = lane < DC::LP && it_ < items;
        const int k = on ? it_ / NW : k_lo, d = on ? it_ - k * NW : 0;
        const double *mytab = tabl + (size_t)(k - k_lo) * TSZL;
        double dxc[NX], acc[NX], dk[NX], dxt[NX], du[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) du[i] = d == i ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) dxc[i] = d == NU + i ? 1.0 : 0.0;
        // The tangent of the ODE at an evaluation point, link by link, with the 9 coefficients of the NEXT link (the next evaluation
        // point's first one after the last; and the link's damping coefficients) requested from LDS before the current link is computed: this wavefront is alone on its
        // SIMD, so nothing else covers the LDS round trip, and with 4 NX doubles of tangents live the compiler placed every read
        // right in front of its use (55 waits on an empty LDS queue per RK4 step: the pass ran on LDS latency).
        constexpr int Mm = M::M;
        double tq[2][12];
        auto fetch = [&](const double *src, double (&t)[12]) {   // a link record: 9 coefficients of the point + the link's damping
            if constexpr (NX <= 21) {
                const d2_t *s2 = (const d2_t *)__builtin_assume_aligned(src, 16);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const d2_t v = s2[j];
                    t[2 * j] = v.x, t[2 * j + 1] = v.y;
                }
            } else {   // (n_mass 6, 7: the lane is out of registers and the aligned register quads of ds_read_b128 cost more than they save: 2.53 vs 2.08 ms)
#pragma unroll
                for (int j = 0; j < 12; ++j) t[j] = src[j];
            }
        };
        auto eval = [&](const double *tb, const double *dxe, auto par0) {   // dk = (d f / d x) dxe + (d f / d u) du; par0: buffer of link 0
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[3 * (Mm + 1) + i] = 0.0;
            static_for<NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, cur = (decltype(par0)::value + i) & 1;
                fetch(tb + (i + 1) * LREC, tq[cur ^ 1]);   // (i + 1 = NL: link 0 of the next evaluation point, the tables are contiguous)
                __builtin_amdgcn_sched_barrier(0);
                M::template ode_tan_link<i>(tq[cur], dxe, du, dk + 3 * (Mm + 1));
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[i] = dxe[3 * (Mm + 1) + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) dk[3 * Mm + j] = du[j];
        };
        constexpr int P1 = NL & 1, P2 = (2 * NL) & 1, P3 = (3 * NL) & 1;   // buffer parity at the start of the evaluation points
        static_assert(((4 * NL) & 1) == 0, "an RK4 step ends on the buffer it started with");
        fetch(mytab, tq[0]);
        for (int s_ = 0; s_ < steps; ++s_) {
            const double *tb = mytab + (size_t)(4 * s_) * NL * LREC;
            eval(tb, dxc, std::integral_constant<int, 0>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + NL * LREC, dxt, std::integral_constant<int, P1>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + 2 * NL * LREC, dxt, std::integral_constant<int, P2>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + h * dk[i];
            eval(tb + 3 * NL * LREC, dxt, std::integral_constant<int, P3>{});
#pragma unroll
            for (int i = 0; i < NX; ++i) dxc[i] = dxc[i] + (h / 6.0) * (acc[i] + dk[i]);
        }
        if (on) {
            double *BA = w + lay.BA + (size_t)k * NX * NW;
#pragma unroll
            for (int i = 0; i < NX; ++i) BA[i * NW + d] = dxc[i];
            if constexpr (ChainCfg<M>::FUSE_GT) {
                // ([B A]_k' nu_{k+1})_d: this lane holds column d.  What the stationarity residual of the round wants (round_start) —
                // as a stage pass over [B A] there it was 40 serial steps through LDS and one more read of the blocks.  Same
                // association as lds_dot<NX> (four partial sums per chunk).
                const double *nun = tabl + DC::CO + M::NTD + (size_t)(k + 1) * NX;
                constexpr int CH = NX <= 12 ? NX : (NX % 12 == 0 ? 12 : (NX % 11 == 0 ? 11 : (NX % 8 == 0 ? 8 : (NX % 7 == 0 ? 7 : 3))));
                double ac[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < NX; ++i) ac[(i % CH) & 3] = fma(dxc[i], nun[i], ac[(i % CH) & 3]);
                w[lay.rt + (size_t)k * NW + d] = (ac[0] + ac[1]) + (ac[2] + ac[3]);
            }
        }
        wave_sync();                                 // the next step's tables overwrite these
    }
}

// ---- the SQP loop of one instance, ONE WAVEFRONT, one launch: per round the linearisation at the current iterate (point pass on N
// lanes, direction pass on all of them), cost / residuals / stopping test, the QP by the Riccati interior-point method, the full step.
// (Until the middle of round 2 every round was a pair of launches and (max_iter + 1) of them were queued per solve: ~43 of the 51
// found nothing to do and cost 19 us of kernel boundaries each.)  The per-round scalars still travel through ws.state, which is
// what the phase functions read.
template <class M>
__global__ void __launch_bounds__(64, 1) chain_sqp_kernel(const LargeSpec sp, const LargeArgs a) {
    using Cfg = ChainCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NT = 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // Cfg::lds_doubles(N) doubles (launch_large)
    __shared__ int sidx[196];
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    ChainSolver<M> S(sp, lane);
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    S.bind_workspace(w, lay);
    S.setup(lds, sidx);
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    const double *x0 = a.x0 + (size_t)inst * NX;
    const double *u0f = S.qmode ? a.u0fix + (size_t)inst * NU : nullptr;
    const int ne = (N + 1) * NW;
    const bool rti = (a.flags & 4) != 0;
    const int max_iter = rti ? 1 : sp.max_iter;
#ifdef MPCRL_PROFILE_PHASES
    __shared__ unsigned long long ph_buf[16];
    S.ph_init(ph_buf);
#endif
    S.ph0();
    int it, n_ipm, status;
    double cost, res[4];
    for (;;) {
        it = (int)S.state[ST_IT];
        n_ipm = (int)S.state[ST_NIPM];
        const bool last_tight = S.state[ST_TIGHT] != 0.0;
        const double stepn = S.state[ST_STEPN];
        // ---- linearisation at the current iterate, cost, NLP residuals
        wave_sync();
        typename ChainSolver<M>::RoundStart rs0;
        if constexpr (MPCRL_CHAIN_MERGE_CALLS != 0 && ChainSolver<M>::USE_V2) {
            rs0 = ChainSolver<M>::round_call(S.ctx(), x0, u0f, sp.h, sp.rk_steps);
        } else {
            for (int e = lane; e < M::NTD; e += NT) lds[Cfg::oBig + DirCfg<M>::CO + e] = S.th[M::td_index(e)];   // (the QP phases reuse the region)
            if constexpr (Cfg::FUSE_GT)
                batched_pass<8>((N + 1) * NX, lane, [&](int e) { return S.NUv[e]; }, [&](int e, double v) { lds[Cfg::oBig + DirCfg<M>::CO + M::NTD + e] = v; });
            wave_sync();
            if (lane < N)
                chain_point_pass<M, false, true>(S.X, S.U, lds + Cfg::oBig + DirCfg<M>::CO, w, N, lane, sp.h, sp.rk_steps,
                                                 lds + Cfg::oBig + DirCfg<M>::CO + M::NTD + (Cfg::FUSE_GT ? (N + 1) * NX : 0));
            wave_sync();
            S.ph(6);
            chain_dir_pass<M>(S.th, w, lds + Cfg::oBig, N, lane, sp.h, sp.rk_steps);
            S.ph(8);
            rs0 = ChainSolver<M>::round_start_call(S.ctx(), x0, u0f);
        }
        cost = rs0.cost;
#pragma unroll
        for (int j = 0; j < 4; ++j) res[j] = rs0.res[j];
        S.ph(9);
        const double rmax = fmax(fmax(res[0], res[1]), fmax(res[2], res[3]));
        status = -1;   // -1: carry on
        if (!(rmax < 1e300) || !(fabs(cost) < 1e300))   // the max-reductions drop NaNs, the cost sum does not
            status = 1;
        else if (rmax < sp.tol && last_tight && !(rti && it == 0))
            status = 0;
        else if (it >= max_iter)
            status = rmax < sp.tol ? 0 : 2;
        if (status >= 0) break;
        const double rr_ = fmin(1.0, rmax), ad_ = rmax < sp.tol ? 0.0 : IPM_ADAPT_C * rr_ * rr_;
        const double tol_res = fmin(IPM_ADAPT_CAP, fmax(IPM_TOL_RES, ad_)), tol_mu = fmin(CHAIN_TOL_MU_FACTOR * IPM_ADAPT_CAP, fmax(IPM_TOL_MU, 1e-2 * ad_));
        const bool tight = tol_res <= IPM_TOL_RES && tol_mu <= IPM_TOL_MU;
        const double warm_mu = stepn < 0.0 ? 0.0 : fmin(IPM_WARM_MAX, fmax(IPM_WARM_MIN, IPM_WARM_C * stepn * stepn));
        // the SQP Hessian: this lane's tiles of (R, Q) without c_k, in registers
        HessConst<M> hs;
        hs.th = S.th, hs.sck = S.sCK();
        // (MPCRL_COLD_DUAL keeps the stored multipliers of the dynamics while the QP starts from zero ones: not the same residual)
        const bool rg_ready = ChainSolver<M>::USE_V2 && MPCRL_CHAIN_SCALE_RES && (stepn >= 0.0 || !(a.flags & 16));
        if (!S.qp_solve(hs, x0, u0f, n_ipm, warm_mu, tol_res, tol_mu, rg_ready)) {
            status = 4;
            break;
        }
        double sl = 0.0;
        batched_pass<4>((N + 1) * NX, lane, [&](int e) { return Quad4{S.dx[e], S.X[e], S.nuq[e], 0.0}; },
                        [&](int e, const Quad4 &v) { sl = fmax(sl, fabs(v.a)), S.X[e] = v.b + v.a, S.NUv[e] = v.c; });
        batched_pass<2>(N * NU, lane, [&](int e) { return Pair2{S.du[e], S.U[e]}; },
                        [&](int e, const Pair2 &v) { sl = fmax(sl, fabs(v.a)), S.U[e] = v.b + v.a; });
        sl = wave_max(sl);
        if (lane == 0) S.state[ST_IT] = it + 1, S.state[ST_NIPM] = n_ipm, S.state[ST_TIGHT] = tight ? 1.0 : 0.0, S.state[ST_STEPN] = sl;
        S.ph(14);
    }
    // ---- finished (converged, failed or out of iterations): results + iterate
    double *PIg = a.PI + (size_t)inst * N * NX;
    const size_t nb = (size_t)(N + 1) * NW;
    double *bnd = a.BND + (size_t)inst * 10 * nb;
    if (lane < NU) a.u0_out[(size_t)inst * NU + lane] = S.U[lane];
    if (lane == 0) {
        a.V[inst] = cost;
        a.status[inst] = status;
        if (a.iters) a.iters[inst * 2] = it, a.iters[inst * 2 + 1] = n_ipm;
        for (int j = 0; j < 4; ++j) a.RES[(size_t)inst * 4 + j] = res[j];
        S.state[ST_ACTIVE] = 0.0, S.state[ST_STATUS] = status;
    }
    // Lagrangian of the mirror, L = cost + pi' g + lam' h (nlp.py:1180; MPC.get_L, mpc.py:325-332): g_k = F(x_k, u_k) - x_{k+1}
    // is the r of this round's linearisation, h = -(slack of the bound row)
    double lag = 0.0;
    for (int e = lane; e < N * NX; e += NT) {
        const double pi_e = S.NUv[NX + e];
        PIg[e] = pi_e;
        lag = fma(pi_e, S.r[e], lag);
    }
    for (int e = lane; e < 2 * ne; e += NT) {
        const int sd = e / ne, ee = e - sd * ne, k = ee / NW, i = ee - k * NW;
        const bool h = !S.skipc(k, i) && S.has(sd, k, i);
        bnd[e] = h ? S.lam[e] : 0.0;
        bnd[2 * nb + e] = h ? S.t[e] : 1.0;
        if (h) lag = fma(-S.lam[e], S.bslack(sd, k, i, S.vc(k, i)), lag);
    }
    lag = wave_sum(lag);
    if (lane == 0 && a.LAG) a.LAG[inst] = cost + lag;
    for (int e = lane; e < 6 * ne; e += NT) bnd[4 * nb + e] = (e >= 4 * ne) ? 1.0 : 0.0;   // no soft rows here
    S.ph_flush();
}

// =====================================================================================================
// sensitivities (dV/dp = dL/dp, nlp.py:1211,1401; du0*/dp by an adjoint Riccati solve, nlp.py:1413-1424), four launches:
//   sens_ad       grid-wide: per (stage, column) one forward-over-reverse sweep of the RK4 map -> exact Lagrangian Hessian blocks;
//                 per stage one reverse sweep -> grad_theta (nu_{k+1}' F_k)
//   sens_riccati  wavefront per instance: dV/dp reductions, factorisation with the exact Hessian + barrier diagonal, nu adjoint solves
//   sens_mix      grid-wide: per (stage, control) the mixed term y_v' d/dv (nu' dF/dtheta) + y_nu' dF/dtheta for all theta at once
//                 (the tangent of the reverse sweep along (y_v, y_nu))
//   sens_out      wavefront per instance: du0*/dp reductions
// They re-use [B A], lam, t, nu left in the workspace by the last SQP round (its linearisation is at the final iterate).
// =====================================================================================================
// grad_theta (nu_{k+1}' F_k): one reverse sweep of the RK4 map per (instance, stage), one lane each.
template <class M>
__global__ void __launch_bounds__(64) chain_sens_th_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD;
    const int N = sp.N;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    const int inst = (int)(gid / N);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const int k = (int)(gid - (long)inst * N);
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU;
    const double *th = a.theta + (size_t)inst * a.theta_stride, *nu = w + lay.ynu;
    double jx[NX], ju[NU], lm[NX], xb[NX], ub[NU], tb[NTD];
    for (int i = 0; i < NU; ++i) ju[i] = U[k * NU + i];
    for (int i = 0; i < NX; ++i) jx[i] = X[k * NX + i], lm[i] = nu[(k + 1) * NX + i];
    disc_map_adj_p<M, true, double>(jx, ju, th, lm, xb, ub, tb, sp.h, sp.rk_steps);
    double *term = w + lay.term + (size_t)k * NTD;
    for (int d = 0; d < NTD; ++d) term[d] = tb[d];
}

// grad_theta (nu_{k+1}' F_k) on the POINT TABLES (when du0*/dp is wanted, chain_point_kernel<M, true> has already walked the adjoint of
// the RK4 map and left, per evaluation point and link, the geometry and the force adjoint q_{e,i}): the parameter adjoint of ode_adj_p
// written out in those quantities — the values of which chain_sens_mix2_kernel forms the tangents.  One lane per (instance, stage), no
// re-evaluation of the map, nothing spilled (the reverse sweep of chain_sens_th_kernel holds four stage vectors and the NTD
// accumulators as one body: 459 / 1 243 spilled registers at n_mass 5 / 7).  The adjoint of the accelerations that the disturbance
// gradient needs is the running sum of the q_{e,i} from the last link down.
template <class M>
__global__ void __launch_bounds__(64) chain_sens_th2_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NTD = M::NTD, NL = M::NL, TAB2 = M::TAB2, MM = M::M;
    const int N = sp.N;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    const int inst = (int)(gid / N);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const int k = (int)(gid - (long)inst * N);
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double *tab = w + lay.ptab + (size_t)k * 8 * NL * TAB2, *qv = w + lay.qvtab + (size_t)k * 8 * NL * 6;
    const double *mp = th, *Dp = th + NL, *Lp = th + 4 * NL;
    double thb[NTD];
#pragma unroll
    for (int d = 0; d < NTD; ++d) thb[d] = 0.0;
    const int ne = 4 * sp.rk_steps;
#pragma unroll 1
    for (int e = 0; e < ne; ++e) {
        double accb[3] = {0.0, 0.0, 0.0};      // adjoint of the acceleration of mass i - 1 while link i is visited (i = NL - 1 .. 1)
#pragma unroll
        for (int i = NL - 1; i >= 0; --i) {
            const double *t = tab + ((size_t)e * NL + i) * TAB2, *q = qv + ((size_t)e * NL + i) * 6, *dv = q + 3;
            const double inrm = sqrt(t[12] * (1.0 / 3.0)), im = 1.0 / mp[i];
            double thm = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double Lj = Lp[3 * i + j], dm = Dp[3 * i + j] * im;
                const double g = 1.0 - Lj * inrm, fd = q[j] * t[j];
                const double gd = im * (fd * g);
                thb[7 * NL + 3 * i + j] += q[j] * dv[j];
                thb[NL + 3 * i + j] += gd;
                thm -= dm * gd;
                thb[4 * NL + 3 * i + j] -= dm * (fd * inrm);
                if (i > 0) {   // q_i = accb_{i-1} - accb_i  (accb_M = 0: the last link ends at the driven mass)
                    accb[j] = q[j] + (i < MM ? accb[j] : 0.0);
                    thb[10 * NL + 3 * (i - 1) + j] += accb[j];
                }
            }
            thb[i] += thm;
        }
    }
    double *term = w + lay.term + (size_t)k * NTD;
#pragma unroll
    for (int d = 0; d < NTD; ++d) term[d] = thb[d];
}

// Exact Lagrangian Hessian of a stage, Hex_k = c_k hess l_k + hess (nu_{k+1}' F_k)(x_k, u_k), ONE WAVEFRONT PER (instance, stage).
// F is the composition of 4 x rk_steps evaluations of the ODE with linear combinations, and the ODE is nonlinear only through the
// spring forces of the links, so the second-order chain rule collapses to
//     hess (nu' F) = sum over evaluation points e and links i of   (d dist_{e,i} / dv)'  G_{e,i}  (d dist_{e,i} / dv),
// G_{e,i} = the 3 x 3 Hessian of (adjoint of the link force at e)' Fs(dist) (ChainDev::link_hessian), d dist / dv = the first-order
// tangents of the link vectors.  Three passes over the evaluation points, each with a quarter of the live state a forward-over-
// reverse jet sweep needs (which spilled ~1000 registers per lane and was bound by its own scratch traffic):
//   1. the point: RK4 in plain doubles; per-link coefficients of every evaluation point        (chain_point_kernel, one lane per stage)
//   2. the adjoint: reverse sweep of nu_{k+1} through the same points; G_{e,i}                    (chain_point_kernel)
//   3. the tangents: lane j < NW carries direction e_j forward; at every evaluation point the wave publishes Y = d dist / dv
//      (3 NL x NW) and W = G Y and accumulates  Hex += Y' W  on the matrix cores: v_mfma_f64_16x16x4, lower tile triangle, operands
//      straight out of LDS in their register layout (A(i, k) and B(k, j) both at lane 16 k + i|j: measured,
//      profiles/microbench/mfma_f64_16x16x4_probe.hip), results D[r](4 r + lane / 16, lane % 16) stored row-coalesced.
template <class M>
struct HexCfg {
    static constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB2 = M::TAB2, EV = 8;
    static constexpr int NTI = (NW + 15) / 16;                 // 16-wide tiles per side
    static constexpr int R = 3 * NL, RP = (R + 3) / 4 * 4;      // rows of Y / W per evaluation point, padded to whole k-steps
    static constexpr int LD = 48;                               // row stride of Y / W (doubles): 384 B, so that the two 32-lane halves
                                                                // of a 64-bit LDS read fall on disjoint bank sets
    static constexpr int oTab = 0, oG = oTab + EV * NL * TAB2, oY = (oG + EV * NL * 6 + 1) & ~1, oW = oY + RP * LD, TOTAL = oW + RP * LD;
    static_assert(NTI * 16 <= LD && NW <= 64, "one direction per lane, tiles inside the padded row");
    // STAGES PER WAVEFRONT (round 4): the tangent propagation keeps NW of the 64 lanes busy and runs on LDS latency (this kernel is
    // one wavefront per SIMD: 4 NX doubles of tangents per lane), so a wavefront takes GW-lane groups of consecutive stages — 2 at
    // n_mass 4-6, 4 at n_mass 3: the same wavefront time for 2 (4) stages.  Each group has its own tables, Y / W and accumulators.
    static constexpr int GW = NW <= 16 ? 16 : (NW <= 32 ? 32 : 64), SPW = 64 / GW;
    __host__ __device__ static constexpr int groups(int N) { return (N + SPW - 1) / SPW; }
    static_assert(SPW * TOTAL * 8 <= 40 * 1024, "four wavefronts per CU");
};

template <class M>
__global__ void __launch_bounds__(64, 1) chain_sens_ad_kernel(const LargeSpec sp, const LargeArgs a) {
    using HC = HexCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NL = M::NL, TAB2 = M::TAB2, LD = HC::LD, RP = HC::RP, NTI = HC::NTI;
    constexpr int GW = HC::GW, SPW = HC::SPW, NTT = NTI * (NTI + 1) / 2;
    typedef double d4_t __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double lds[SPW * HC::TOTAL];
    const int N = sp.N, lane = threadIdx.x, NG = HC::groups(N);
    const int inst = blockIdx.x / NG, k0 = (blockIdx.x - inst * NG) * SPW;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double h = sp.h;
    const int steps = sp.rk_steps;
    const int sub = lane / GW, dl = lane - sub * GW;      // this lane's stage group and direction
    double *const mine = lds + sub * HC::TOTAL;
    const double *tab = mine + HC::oTab;
    for (int s_ = 0; s_ < SPW; ++s_) {   // 1. + 2. the point and the adjoint were done per stage by chain_point_kernel<M, true>: its tables -> LDS
        const int k = min(k0 + s_, N - 1);   // (a group past the horizon repeats the last stage and is not stored)
        const double *pt = w + lay.ptab + (size_t)k * HC::EV * NL * TAB2, *gt = w + lay.gtab + (size_t)k * HC::EV * NL * 6;
        double *tb_ = lds + s_ * HC::TOTAL;
        for (int e = lane; e < HC::EV * NL * TAB2; e += 64) tb_[HC::oTab + e] = pt[e];
        for (int e = lane; e < HC::EV * NL * 6; e += 64) tb_[HC::oG + e] = gt[e];
        for (int r = lane; r < RP * LD; r += 64) tb_[HC::oY + r] = 0.0, tb_[HC::oW + r] = 0.0;   // padding rows / columns stay zero
    }
    wave_sync();
    // 3. the tangents and the Hessian accumulation
    d4_t D[SPW][NTT];
#pragma unroll
    for (int s_ = 0; s_ < SPW; ++s_)
#pragma unroll
        for (int t_ = 0; t_ < NTT; ++t_) D[s_][t_] = d4_t{0.0, 0.0, 0.0, 0.0};
    {
        double dxc[NX], acc[NX], dk[NX], dxt[NX], du[NU], dd[3 * NL];
#pragma unroll
        for (int i = 0; i < NU; ++i) du[i] = dl == i ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) dxc[i] = dl == NU + i ? 1.0 : 0.0;      // lanes >= NW of a group carry the zero direction
        const int lr = lane >> 4, lc = lane & 15;
        auto accumulate = [&](int e) {   // publish this evaluation point's Y, W = G Y and add Y' W to the tiles
            const double *G = mine + HC::oG + (size_t)e * NL * 6;
            double *Y = mine + HC::oY, *W = mine + HC::oW;
            if (dl < NW) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const double *g = G + i * 6, d0 = dd[3 * i], d1 = dd[3 * i + 1], d2 = dd[3 * i + 2];
                    Y[(3 * i) * LD + dl] = d0, Y[(3 * i + 1) * LD + dl] = d1, Y[(3 * i + 2) * LD + dl] = d2;
                    W[(3 * i) * LD + dl] = g[0] * d0 + g[1] * d1 + g[3] * d2;
                    W[(3 * i + 1) * LD + dl] = g[1] * d0 + g[2] * d1 + g[4] * d2;
                    W[(3 * i + 2) * LD + dl] = g[3] * d0 + g[4] * d1 + g[5] * d2;
                }
            }
            wave_sync();
#pragma unroll
            for (int s_ = 0; s_ < SPW; ++s_) {
                const double *Ys = lds + s_ * HC::TOTAL + HC::oY, *Ws = lds + s_ * HC::TOTAL + HC::oW;
#pragma unroll
                for (int ks = 0; ks < RP / 4; ++ks) {
                    double ya[NTI], wb[NTI];
#pragma unroll
                    for (int t_ = 0; t_ < NTI; ++t_) ya[t_] = Ys[(4 * ks + lr) * LD + 16 * t_ + lc], wb[t_] = Ws[(4 * ks + lr) * LD + 16 * t_ + lc];
#pragma unroll
                    for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj)
                            D[s_][ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ti], wb[tj], D[s_][ti * (ti + 1) / 2 + tj], 0, 0, 0);
                }
            }
            wave_sync();
        };
        // the tangent of the ODE link by link, the coefficients of the next link in flight while this one is computed (as in
        // chain_dir_pass: one wavefront per SIMD, nothing else covers an LDS round trip)
        constexpr int Mm = M::M;
        double tq[2][12];
        double Cr[3 * NL];
#pragma unroll
        for (int i = 0; i < 3 * NL; ++i) Cr[i] = th[7 * NL + i];
        auto fetch = [&](const double *src, double (&t)[12]) {
#pragma unroll
            for (int j = 0; j < 9; ++j) t[j] = src[j];
        };
        auto eval = [&](const double *tb, const double *dxe, auto par0) {
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[3 * (Mm + 1) + i] = 0.0;
            static_for<NL>([&](auto i_) {
                constexpr int i = decltype(i_)::value, cur = (decltype(par0)::value + i) & 1;
                fetch(tb + (i + 1) * TAB2, tq[cur ^ 1]);     // (i + 1 = NL: link 0 of the next evaluation point, the tables are contiguous)
#pragma unroll
                for (int j = 0; j < 3; ++j) tq[cur][9 + j] = Cr[3 * i + j];
                __builtin_amdgcn_sched_barrier(0);
                M::template ode_tan_link<i, true>(tq[cur], dxe, du, dk + 3 * (Mm + 1), dd);
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int i = 0; i < 3 * Mm; ++i) dk[i] = dxe[3 * (Mm + 1) + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) dk[3 * Mm + j] = du[j];
        };
        constexpr int P1 = NL & 1, P2 = (2 * NL) & 1, P3 = (3 * NL) & 1;
        fetch(tab, tq[0]);
        for (int s = 0; s < steps; ++s) {
            const double *tb = tab + (size_t)(4 * s) * NL * TAB2;
            eval(tb, dxc, std::integral_constant<int, 0>{});
            accumulate(4 * s);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + NL * TAB2, dxt, std::integral_constant<int, P1>{});
            accumulate(4 * s + 1);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + (0.5 * h) * dk[i];
            eval(tb + 2 * NL * TAB2, dxt, std::integral_constant<int, P2>{});
            accumulate(4 * s + 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = acc[i] + 2.0 * dk[i], dxt[i] = dxc[i] + h * dk[i];
            eval(tb + 3 * NL * TAB2, dxt, std::integral_constant<int, P3>{});
            accumulate(4 * s + 3);
#pragma unroll
            for (int i = 0; i < NX; ++i) dxc[i] = dxc[i] + (h / 6.0) * (acc[i] + dk[i]);
        }
        // Hex_k = c_k hess l_k + the accumulated second-order term; D[.][r] holds (row 4 r + lane / 16, column lane % 16) of its tile
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            const int k = k0 + s_;
            if (k >= N) break;
            const double ckk = sp.cost_kind == 0 ? sp.dT : (k == 0 ? sp.dT : pow(sp.gamma, (double)k) * sp.dT);
            double *Hex = w + lay.Hex + (size_t)k * NW * NW;
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ti + 4 * r + lr, j = 16 * tj + lc;
                        if (i < NW && j < NW) {
                            const double v = fma(ckk, M::hess(false, i, j, th), D[s_][ti * (ti + 1) / 2 + tj][r]);
                            Hex[i * NW + j] = v;
                            if (ti != tj) Hex[j * NW + i] = v;
                        }
                    }
        }
    }
}

template <class M>
__global__ void __launch_bounds__(64, 1) chain_sens_riccati_kernel(const LargeSpec sp, const LargeArgs a) {
    using Cfg = ChainCfg<M>;
    constexpr int NX = M::NX, NU = M::NU, NW = NX + NU, NTD = M::NTD, NP = M::NP, NT = 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // Cfg::lds_doubles(N) doubles (launch_large)
    __shared__ int sidx[196];
    const int lane = threadIdx.x, inst = blockIdx.x, N = sp.N;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    ChainSolver<M> S(sp, lane);
    S.th = a.theta + (size_t)inst * a.theta_stride;
    S.qmode = a.u0fix != nullptr;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    S.bind_workspace(w, lay);
    S.setup(lds, sidx);
    S.X = a.X + (size_t)inst * (N + 1) * NX, S.U = a.U + (size_t)inst * N * NU;
    const int ne = (N + 1) * NW;
    const double *th = S.th, *xs = sp.consts;
    const WsArr Hex{(char *)w, (unsigned)lay.Hex};
    if (!((a.flags & 2) && a.dpi) || S.qmode) return;
    for (int e = lane; e < NW * NW; e += NT) Hex[N * NW * NW + e] = S.ck(N) * M::hess(true, e / NW, e % NW, th);
    // barrier diagonal from the final (lam, t) of the bound rows (slacks are constants of the mirror, quirk q1)
    for (int e = lane; e < ne; e += NT) {
        const int k = e / NW, i = e - k * NW;
        double d = 0.0;
        if (!S.skipc(k, i))
            for (int sd = 0; sd < 2; ++sd)
                if (S.has(sd, k, i)) d += S.LAM(sd, e) / S.TT(sd, e);
        S.Dg[e] = d;
    }
    wave_sync();
    HessGlobal<M> hs;
    hs.Hex = Hex;
    for (int e = lane; e < N * NX; e += NT) S.rb[e] = 0.0;   // no dynamics offset in the adjoint systems
    const WsArr Ydx{(char *)w, (unsigned)lay.Ydx}, Ydu{(char *)w, (unsigned)lay.Ydu}, Ydnu{(char *)w, (unsigned)lay.Ydnu};
    bool okall = true;
    if constexpr (ChainSolver<M>::USE_V2_SENS) {      // one factor sweep, ONE forward sweep for the NU adjoint solves
        for (int e = lane; e < ne; e += NT) S.rt[e] = e == 0 ? -1.0 : 0.0;
        wave_sync();
        okall = ChainSolver<M>::template factor_call<HessGlobal<M>>(S.ctx(), hs.hex_offset(), S.rt.off, S.rb.off);
        ChainSolver<M>::forward_sens_call(S.ctx(), Ydx.off, Ydu.off, Ydnu.off);
    } else
    for (int iu = 0; iu < NU; ++iu) {
        for (int e = lane; e < ne; e += NT) S.rt[e] = e == iu ? -1.0 : 0.0;
        wave_sync();
        if (iu == 0)
            okall = ChainSolver<M>::template factor_call<HessGlobal<M>>(S.ctx(), hs.hex_offset(), S.rt.off, S.rb.off);
        else if (lane == 0) {
            // The right-hand side is -e_iu in the controls of stage 0 and zero elsewhere, and there is no dynamics offset: the backward
            // vector recursion is identically zero from the terminal stage down to stage 1 (p_k = 0 — what the factor sweep left for
            // iu = 0 as well), and at stage 0 only the feed-forward changes: kff_0 = (L_0 L_0')^{-1} (-e_iu).  No sweep.
            if constexpr (ChainSolver<M>::USE_V2_SENS) {
#pragma unroll
                for (int i = 0; i < NU; ++i) S.kff[i] = -S.minv2[4 * i + iu];
            } else {
                double y[NU], z[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    double a_ = i == iu ? -1.0 : 0.0;
#pragma unroll
                    for (int m = 0; m < i; ++m) a_ -= S.L[i * NU + m] * y[m];
                    y[i] = a_ * S.L[i * NU + i];
                }
#pragma unroll
                for (int i = NU - 1; i >= 0; --i) {
                    double a_ = y[i];
#pragma unroll
                    for (int m = i + 1; m < NU; ++m) a_ -= S.L[m * NU + i] * z[m];
                    z[i] = a_ * S.L[i * NU + i];
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) S.kff[i] = z[i];
            }
        }
        wave_sync();
        ChainSolver<M>::template forward_call<true>(S.ctx(), S.rb.off);
        for (int e = lane; e < (N + 1) * NX; e += NT) Ydx[iu * (N + 1) * NX + e] = S.Dx[e], Ydnu[iu * (N + 1) * NX + e] = S.Dnu[e];
        for (int e = lane; e < N * NU; e += NT) Ydu[iu * N * NU + e] = S.Du[e];
        wave_sync();
    }
    if (lane == 0) S.state[ST_STATUS] = okall ? 0.0 : 4.0;   // read by sens_out: NaN sensitivities when the exact-Hessian KKT matrix is not pd
}

template <class M>
__global__ void __launch_bounds__(256) chain_sens_mix_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD;
    const int N = sp.N;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int per = N * NU;
    const int inst = (int)(gid / per);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const int it = (int)(gid - (long)inst * per), iu = it / N, k = it - iu * N;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU;
    const double *th = a.theta + (size_t)inst * a.theta_stride, *nu = w + lay.ynu;
    const double *Ydx = w + lay.Ydx + (size_t)iu * (N + 1) * NX, *Ydu = w + lay.Ydu + (size_t)iu * N * NU, *Ydnu = w + lay.Ydnu + (size_t)iu * (N + 1) * NX;
    Jet1<1> jx[NX], ju[NU], lm[NX], xb[NX], ub[NU], tb[NTD];
    for (int c = 0; c < NU; ++c) ju[c] = Jet1<1>(U[k * NU + c]), ju[c].d[0] = Ydu[k * NU + c];
    for (int c = 0; c < NX; ++c) {
        jx[c] = Jet1<1>(X[k * NX + c]), jx[c].d[0] = Ydx[k * NX + c];
        lm[c] = Jet1<1>(nu[(k + 1) * NX + c]), lm[c].d[0] = Ydnu[(k + 1) * NX + c];
    }
    disc_map_adj_p<M, true, Jet1<1>>(jx, ju, th, lm, xb, ub, tb, sp.h, sp.rk_steps);
    double *term2 = w + lay.term2 + ((size_t)iu * N + k) * NTD;
    for (int d = 0; d < NTD; ++d) term2[d] = tb[d].d[0];
}

// ---- the mixed term on the POINT TABLES (round 4; replaces chain_sens_mix_kernel, whose forward-over-reverse jet sweep of the whole
// RK4 map kept four stage states, four adjoint vectors and the parameter adjoint live as jets: 1 076 spilled registers and 3.3 KB of
// scratch per lane at n_mass 5, 1 696 / 6 KB at n_mass 7 — 0.43 / 1.85 ms for 12 k flops per lane).  Per (instance, stage, control)
//     term2 = d/d eps  grad_theta [nu' F](v + eps y_v, nu + eps y_nu)
// F is RK4 steps of an ODE whose only nonlinearity is the spring force of each link, so everything second order is local to a
// (evaluation point e, link i): with the point tables of chain_point_pass<true> (dist, spring coefficients; G_{e,i}; the force
// adjoint q_{e,i} and the velocity difference dv_{e,i}) the sweep is
//   1. the tangent of the evaluation states along y_v: plain ode_tan on the tables (what the direction pass does), one RK4 step at a time;
//   2. the tangent of the reverse sweep: the same linear recursion as the values (ode_tan_T) on the tangent adjoints, plus the source
//      (d dist / dx)' G_{e,i} d dist_{e,i} at every evaluation point;
//   3. at every (e, i) the tangent of the parameter adjoint of ode_adj_p, written out in the table quantities.
// ~170 doubles of live state, no jets, no scratch.
template <class M>
__global__ void __launch_bounds__(64) chain_sens_mix2_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD, NL = M::NL, TAB2 = M::TAB2, MM = M::M;
    // per lane in LDS ([index][lane]: conflict-free): the NTD accumulators of the parameter-adjoint tangent and the start state of
    // the second RK4 step — with the evaluation states recomputed where they are used, what stays in registers is one evaluation
    // state, one ODE tangent and the four adjoint vectors
    extern __shared__ double sm[];
    const int N = sp.N, lane = threadIdx.x;
    double *thd = sm + lane, *dx1 = sm + NTD * 64 + lane;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    const int per = N * NU;
    const int inst = (int)(gid / per);
    if (inst >= a.B) return;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    // (the control index runs fastest: the NU lanes of a stage read the SAME table entries — every table access of this kernel is a gather,
    // one cache line per lane, and the kernel runs on the rate of those requests; with the stage index fastest all 64 lanes of a load
    // went to different lines)
    const int it = (int)(gid - (long)inst * per), k = it / NU, iu = it - k * NU;
    const LargeLayout<M> lay(N);
    double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *th = a.theta + (size_t)inst * a.theta_stride;
    const double *tab = w + lay.ptab + (size_t)k * 8 * NL * TAB2, *Gt = w + lay.gtab + (size_t)k * 8 * NL * 6, *qv = w + lay.qvtab + (size_t)k * 8 * NL * 6;
    const double *Ydx = w + lay.Ydx + (size_t)iu * (N + 1) * NX + (size_t)k * NX, *Ydu = w + lay.Ydu + (size_t)iu * N * NU,
                 *Ydnu = w + lay.Ydnu + (size_t)iu * (N + 1) * NX;
    const double h = sp.h;
    const int steps = sp.rk_steps;
    const double *mp = th, *Dp = th + NL, *Lp = th + 4 * NL;
    double du[NU], lbd[NX];
#pragma unroll
    for (int c = 0; c < NU; ++c) du[c] = Ydu[k * NU + c];
#pragma unroll
    for (int c = 0; c < NX; ++c) lbd[c] = Ydnu[(k + 1) * NX + c];
#pragma unroll
    for (int d = 0; d < NTD; ++d) thd[d * 64] = 0.0;
    auto dx0 = [&](int s, int i) { return s == 0 ? Ydx[i] : dx1[i * 64]; };      // start state of step s
    // evaluation state n (0..3) of step s into dX.  Runtime loops on purpose (no unrolling over the evaluation points): straight-line
    // code over the eight points lets the scheduler stretch live ranges over all of them, and the kernel is back in scratch
    auto state_at = [&](int s, int n, double (&dX)[NX]) {
        const double *tb = tab + (size_t)(4 * s) * NL * TAB2;
#pragma unroll
        for (int i = 0; i < NX; ++i) dX[i] = dx0(s, i);
#pragma unroll 1
        for (int m = 0; m < n; ++m) {
            double dk[NX];
            M::template ode_tan<TAB2, false>(tb + (size_t)m * NL * TAB2, th, dX, du, dk, nullptr);
            const double c = m == 2 ? h : 0.5 * h;
#pragma unroll
            for (int i = 0; i < NX; ++i) dX[i] = dx0(s, i) + c * dk[i];
        }
    };
    if (steps > 1) {      // start state of the second step: one full RK4 step of the tangent
        const double *tb = tab;
        double dX[NX], acc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dX[i] = dx0(0, i), acc[i] = 0.0;
#pragma unroll 1
        for (int m = 0; m < 4; ++m) {
            double dk[NX];
            M::template ode_tan<TAB2, false>(tb + (size_t)m * NL * TAB2, th, dX, du, dk, nullptr);
            const double c = m == 2 ? h : 0.5 * h, wgt = (m == 0 || m == 3) ? 1.0 : 2.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = fma(wgt, dk[i], acc[i]), dX[i] = dx0(0, i) + c * dk[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) dx1[i * 64] = dx0(0, i) + (h / 6.0) * acc[i];
    }
#pragma unroll 1
    for (int s = steps - 1; s >= 0; --s) {
        double kbd[NX], Xbd[NX], accd[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) accd[i] = lbd[i], Xbd[i] = 0.0;
#pragma unroll 1
        for (int n = 3; n >= 0; --n) {      // evaluation point e = 4 s + n: Xbd = J' kbd + Hessian source; parameter-adjoint tangents
            // weights of the reverse RK4 sweep: kb_3 = h/6 lb, kb_2 = h/3 lb + h Xb_3, kb_1 = h/3 lb + h/2 Xb_2, kb_0 = h/6 lb + h/2 Xb_1
            const double ca = (n == 3 || n == 0) ? h / 6.0 : h / 3.0, cb = n == 3 ? 0.0 : (n == 2 ? h : 0.5 * h);
#pragma unroll
            for (int i = 0; i < NX; ++i) kbd[i] = ca * lbd[i] + cb * Xbd[i];
            const int e = 4 * s + n;
            const double *tb = tab + (size_t)e * NL * TAB2;
            double qd[3 * NL], dX[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) Xbd[i] = 0.0;
            M::template ode_tan_T<TAB2>(tb, th, kbd, Xbd, qd);
#pragma unroll
            for (int i = 0; i < 3 * MM; ++i) thd[(10 * NL + i) * 64] += kbd[3 * (MM + 1) + i];        // w enters the accelerations directly
            state_at(s, n, dX);
            const double *dpos = dX, *dvel = dX + 3 * (MM + 1);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const double *t = tb + i * TAB2, *G = Gt + ((size_t)e * NL + i) * 6, *q = qv + ((size_t)e * NL + i) * 6, *dv = q + 3;
                double dd[3], ddv[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    dd[j] = i ? dpos[3 * i + j] - dpos[3 * (i > 0 ? i - 1 : 0) + j] : dpos[j];
                    const double dvr = i < MM ? dvel[3 * (i < MM ? i : 0) + j] : du[j];
                    ddv[j] = i ? dvr - dvel[3 * (i > 0 ? i - 1 : 0) + j] : dvr;
                }
                const double w0 = G[0] * dd[0] + G[1] * dd[1] + G[3] * dd[2], w1 = G[1] * dd[0] + G[2] * dd[1] + G[4] * dd[2],
                             w2 = G[3] * dd[0] + G[4] * dd[1] + G[5] * dd[2];
                const double wv[3] = {w0, w1, w2};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Xbd[3 * i + j] += wv[j];
                    if (i > 0) Xbd[3 * (i > 0 ? i - 1 : 0) + j] -= wv[j];
                }
                const double inrm = sqrt(t[12] * (1.0 / 3.0)), sdot = t[0] * dd[0] + t[1] * dd[1] + t[2] * dd[2];
                const double dinrm = -(inrm * inrm * inrm) * sdot, im = 1.0 / mp[i];
                double thm = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double Lj = Lp[3 * i + j], dm = Dp[3 * i + j] * im;
                    const double g = 1.0 - Lj * inrm, dg = -Lj * dinrm;
                    const double fd = q[j] * t[j], dfd = qd[3 * i + j] * t[j] + q[j] * dd[j];
                    const double dgd = im * (dfd * g + fd * dg);
                    thd[(7 * NL + 3 * i + j) * 64] += qd[3 * i + j] * dv[j] + q[j] * ddv[j];
                    thd[(NL + 3 * i + j) * 64] += dgd;
                    thm -= dm * dgd;
                    thd[(4 * NL + 3 * i + j) * 64] -= dm * (dfd * inrm + fd * dinrm);
                }
                thd[i * 64] += thm;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) accd[i] += Xbd[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) lbd[i] = accd[i];
    }
    double *term2 = w + lay.term2 + ((size_t)iu * N + k) * NTD;
#pragma unroll
    for (int d = 0; d < NTD; ++d) term2[d] = thd[d * 64];
}

// One output element per lane: slot 0 = dV/dp (with MPCRL_SENS_V), slots 1..NU = rows of du0*/dp (with MPCRL_SENS_PI).
// Each element is a sum over the stages of per-stage terms left in the workspace, or of closed forms in (X, U, adjoint solution).
template <class M>
__global__ void __launch_bounds__(1024) chain_sens_out_kernel(const LargeSpec sp, const LargeArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NTD = M::NTD, NP = M::NP;
    constexpr int NE = NTD + NX * NX + NU * NU;   // elements per slot: dynamics parameters, Q (column-major), R
    constexpr int PER = (NU + 1) * NE;
    const int N = sp.N;
    // One workgroup of 1024 lanes per INSTANCE.  The Q / R outputs are sums over the stages of products of two trajectory entries:
    // the trajectories (X - x_ss, U, the NU adjoint solutions: 32 KB at n_mass 5) are staged in LDS once, coalesced, and every lane
    // then takes outputs rem = lane, lane + 1024, ...  (Round 3 ran one lane per output on 256-lane workgroups that each went to
    // global memory with 64 different addresses per load instruction: 0.22 ms; staged per 256 outputs: 0.20 ms — the staging latency
    // of 8 workgroups per instance, 6 rounds of them on the chip, was the time.)
    extern __shared__ double sm[];
    double *cks = sm, *lX = sm + 64, *lU = lX + (N + 1) * NX, *lY = lU + N * NU;   // lY: [NU][(N+1) NX + N NU]
    const int inst = blockIdx.x, tid = threadIdx.x;
    const int status = a.status[inst];
    if (!(status == 0 || status == 2)) return;
    const bool want_v = (a.flags & 1) && a.dV, want_pi = (a.flags & 2) && a.dpi && !a.u0fix;
    const LargeLayout<M> lay(N);
    const double *w = a.ws + (size_t)inst * a.ws_stride;
    const double *X = a.X + (size_t)inst * (N + 1) * NX, *U = a.U + (size_t)inst * N * NU, *xs = sp.consts;
    const int ny = (N + 1) * NX + N * NU;
    for (int k = tid; k <= N; k += 1024) {
        double c = k == N ? 1.0 : sp.dT;
        if (sp.cost_kind != 0) c = k == 0 ? sp.dT : (k == N ? pow(sp.gamma, (double)N) : pow(sp.gamma, (double)k) * sp.dT);
        cks[k] = c;
    }
    for (int e = tid; e < (N + 1) * NX; e += 1024) lX[e] = X[e] - xs[e % NX];
    for (int e = tid; e < N * NU; e += 1024) lU[e] = U[e];
    if (want_pi)
        for (int e = tid; e < NU * ny; e += 1024) {
            const int iu = e / ny, o = e - iu * ny;
            lY[e] = o < (N + 1) * NX ? w[lay.Ydx + (size_t)iu * (N + 1) * NX + o] : w[lay.Ydu + (size_t)iu * N * NU + (o - (N + 1) * NX)];
        }
    __syncthreads();
    const bool sens_ok = w[lay.state + ST_STATUS] == 0.0;
    for (int rem = tid; rem < PER; rem += 1024) {
        const int slot = rem / NE, e0 = rem - slot * NE;
        if (slot == 0 ? !want_v : !want_pi) continue;
        const int iu = slot - 1;
        const double *tm = slot == 0 ? w + lay.term : w + lay.term2 + (size_t)iu * N * NTD;
        const double *Dx = lY + (iu < 0 ? 0 : iu) * ny, *Du = Dx + (N + 1) * NX;
        double acc = 0.0;
        int pidx;
        if (e0 < NTD) {
            for (int k = 0; k < N; ++k) acc += tm[k * NTD + e0];
            pidx = M::td_index(e0);
        } else if (e0 < NTD + NX * NX) {
            const int e = e0 - NTD, j = e / NX, i = e - j * NX;   // column-major position of Q(i, j)
            if (slot == 0) {   // d/dQ_ij of sum_k c_k l_k (ocp_utils.py:276-277)
                for (int k = 0; k <= N; ++k) acc = fma(0.5 * cks[k] * lX[k * NX + i], lX[k * NX + j], acc);
            } else {           // y' d2 l / dv dQ_ij = 1/2 (y_i e_j + y_j e_i)
                for (int k = 0; k <= N; ++k) acc += 0.5 * cks[k] * (Dx[k * NX + i] * lX[k * NX + j] + Dx[k * NX + j] * lX[k * NX + i]);
            }
            pidx = M::OFF_Q + e;
        } else {
            const int ee = e0 - NTD - NX * NX, j = ee / NU, i = ee - j * NU;
            if (slot == 0) {
                for (int k = 0; k < N; ++k) acc = fma(0.5 * cks[k] * lU[k * NU + i], lU[k * NU + j], acc);
            } else {
                for (int k = 0; k < N; ++k) acc += 0.5 * cks[k] * (Du[k * NU + i] * lU[k * NU + j] + Du[k * NU + j] * lU[k * NU + i]);
            }
            pidx = M::OFF_R + ee;
        }
        if (slot == 0)
            a.dV[(size_t)inst * NP + pidx] = acc;
        else
            a.dpi[((size_t)inst * NU + iu) * NP + pidx] = sens_ok ? -acc : NAN;
    }
}

}  // namespace mpcrl
