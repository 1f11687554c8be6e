/* mpcrl.h — C ABI of the MI355X-native batched MPC-as-policy engine (libmpcrl_hip.so).
 *
 * This is the drop-in boundary for the hot path of MPC-Based-Reinforcement-Learning/mpc4rl:
 *
 *   reference interface replaced                               entry point here
 *   --------------------------------------------------------   ---------------------------------
 *   AcadosOcpSolver(ocp) + build_nlp(ocp)                      mpcrl_create
 *     rlmpc/mpc/cartpole/acados.py:192-203,
 *     rlmpc/mpc/linear_system/acados.py:20-22,126-129,
 *     rlmpc/mpc/chain_mass/acados.py:27-29,45
 *   ocp_solver.set(stage,"p",..) / cost_set(stage,"W"|"yref")  mpcrl_set_theta   (the cartpole cost block W_0, W, W_e,
 *     rlmpc/mpc/common/mpc.py:137-154,212-257                    yref_0, yref, yref_e of p is what the solve uses)
 *   ocp_solver.constraints_set(stage,"lbu"|"ubu"|"lbx"|"ubx")  mpcrl_set_bounds
 *     rlmpc/mpc/common/mpc.py:72-73,87-88
 *   float(nlp.L.val)  (MPC.get_L)                              mpcrl_get_lagrangian
 *     rlmpc/mpc/common/mpc.py:325-332
 *   shared_lib.ocp_nlp_cost_model_set(.., "scaling", gamma^k)  mpcrl_set_gamma
 *     rlmpc/mpc/common/mpc.py:259-285  (the reference's one raw C call)
 *   ocp_solver.reset(); set(stage,"x",x0)                      mpcrl_reset
 *     rlmpc/mpc/common/mpc.py:204-210
 *   set(0,"lbx"/"ubx",x0) [+ constraints_set(0,"lbu"/"ubu",u0)];   mpcrl_solve
 *   ocp_solver.solve(); get(0,"u"); get_cost(); update_nlp()
 *     rlmpc/mpc/common/mpc.py:27-50,52-96,177-202 and
 *     rlmpc/mpc/nlp.py:1341-1424 (dL_dp, dpi_dp)
 *   ocp_solver.get(stage, "x"|"u"|"pi"|"lam"|"t"|"sl"|"su")   mpcrl_get_iterate
 *   ocp_solver.set(stage, "x"|"u"|"pi", v) / load_iterate      mpcrl_set_iterate
 *     rlmpc/mpc/nlp.py:1354-1372, rlmpc/examples/chain_mass.py:119-120
 *   the one warm solver object SB3's replay loop reuses          mpcrl_get_iterate_rows / mpcrl_set_iterate_rows
 *     rlmpc/td3/policies.py:186-213                              (per-transition iterates of a replay buffer)
 *
 * Conventions
 *   - plain C, no torch types.  Every array argument of mpcrl_solve / *_iterate / mpcrl_reset /
 *     mpcrl_set_theta is a DEVICE pointer (HIP), double precision, row-major, owned by the caller.
 *   - the handle owns the warm-start iterate and all workspace; mpcrl_solve allocates nothing and is
 *     asynchronous on the given HIP stream.  A new handle holds the cold iterate (x = 0, u = 0, multipliers 0,
 *     slacks 1); mpcrl_reset(h, x0) restores it with x_k = x0.
 *   - return value: 0 ok, < 0 API misuse (MPCRL_E_*).  Per-instance solver status goes to status[B]
 *     with acados' numbering: 0 success, 1 NaN, 2 max-iter, 4 QP failure (reference: solve() returns
 *     int status, update/q_update raise on != 0, rlmpc/mpc/common/mpc.py:81-83,197-198).
 *   - one handle per (device, stream); handles are independent across GPUs; not re-entrant (stateful
 *     warm start, like the reference's solver object).
 *   - how a batch is laid out over wavefronts (packing order, plain or time-sliced launch of the small
 *     solve kernel; environment MPCRL_TIME_SLICE=0/1 read at mpcrl_create overrides the automatic choice)
 *     never changes a result bit.
 *   - linear-system model: environment MPCRL_LINEAR_SPL=1 at mpcrl_create keeps the one-stage-per-lane solve kernel instead
 *     of the three-stages-per-lane one.  Same iteration, sums associated differently: results equal to rounding EXCEPT where a
 *     stopping test is met within rounding — there the interior-point count can differ by one, an RTI call's status can differ,
 *     and the outputs agree to the QP tolerance only (see ABI 110 below).
 *   - linear-system model, warm calls: an instance whose WARM-started interior point runs out of iterations (it can jam against the
 *     rows a moved x0 activates) solves that QP once more from the cold interior point before status 4 is reported — the QP is convex,
 *     the cold start solves it; its iteration count then includes both attempts (> 60).  No signature or output layout changes.
 *   - cartpole: a wavefront holds min(floor(64 / (N + 1)), 4) instances — four is the number of 4x4 blocks of the
 *     matrix-core sweeps — so horizons below N = 15 use fewer of its lanes than a lane-per-stage packing could.
 */
#ifndef MPCRL_H
#define MPCRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = what mpcrl_version() of the library returns.  Bumped whenever the signature or the meaning of an exported function
 * changes; a binding must compare the two before its first call (mpc4rl_amd/_lib.py does).
 *   100  rounds 1-3
 *   110  round 4/5: mpcrl_query_time_sliced(h, flags, stream) gained `stream`; mpcrl_env_cartpole_step / _reset and
 *        mpcrl_env_linear_step gained `obs_f32` (obs became void*); mpcrl_set_exit_rule accepts the chain of masses;
 *        mpcrl_create refuses chain horizons whose trajectories do not fit the LDS of one workgroup (MPCRL_E_ARG).
 *        Behavioural change under 110 (round 5): the linear-system model's default solve kernel became lq_solve_kernel (three
 *        stages per lane; MPCRL_LINEAR_SPL=1 keeps the old one) — the SAME iteration with its sums associated differently, so a
 *        stopping test that is met within rounding can end an interior-point loop one iteration earlier or later (< 1 % of
 *        instances), or flip the status of an RTI call whose residual sits at the tolerance; outputs of such an instance then
 *        agree with the one-stage kernel and the oracle port to the QP tolerance (1e-4 ... 1e-3 on du0/dp), not to rounding
 *   120  round 6: mpcrl_solve flags MPCRL_NO_BND_STORE and MPCRL_EXACT_QP (test-only); mpcrl_get_iterate_rows / mpcrl_set_iterate_rows; mpcrl_policy_action
 *   130  round 6: mpcrl_critic_td_grad / mpcrl_critic_workspace_bytes / mpcrl_critic_dq_da (the TD3 learner's critic step); mpcrl_replay_sample; mpcrl_dpg_grad / mpcrl_dpg_workspace_bytes; mpcrl_td3_cartpole_collect; mpcrl_td3_policy_post; linear system: a
 *        failed WARM QP restarts cold (behaviour, see above) */
#define MPCRL_ABI_VERSION 130

enum { MPCRL_MODEL_CARTPOLE = 0, MPCRL_MODEL_LINEAR = 1, MPCRL_MODEL_CHAIN = 2 };
/* how the stage-cost scaling c_k is built (rlmpc/mpc/nlp.py:1044-1055 vs 1083-1091) */
enum { MPCRL_COST_NLS = 0, MPCRL_COST_EXTERNAL = 1 };
/* mpcrl_solve flags */
enum {
    MPCRL_SENS_V = 1,  /* dV/dp (or dQ/dp with u0_fixed)  = dL/dp   nlp.py:1211,1401 */
    MPCRL_SENS_PI = 2, /* du0* / dp                        nlp.py:1413-1424 */
    MPCRL_RTI = 4,     /* one SQP iteration from the stored iterate (build-side mode; the reference always runs full SQP) */
    MPCRL_COLD = 8,    /* ignore the stored iterate: x_k = x0, u = 0, multipliers 0 (MPC.reset, mpc.py:204-210) */
    MPCRL_COLD_DUAL = 16, /* start from the stored x, u, pi but ignore the stored bound multipliers / slacks (the interior point
                            starts from its default point).  Implied for the first solve after mpcrl_set_iterate(bnd = NULL):
                            the reference's `for stage: ocp_solver.set(stage, "x", x0)` initial guess (mpc.py:208-210) */
    MPCRL_NO_BND_STORE = 32, /* ABI 120.  Do not write the bound multipliers / slacks of the solution back into the handle's stored
                            iterate (ten planes of (N+1)(nx+nu) doubles per instance: 3.4 of the 4.7 KB a linear-system solve
                            writes).  x, u, pi are still stored; the planes keep what they held, the NEXT solve behaves as with
                            MPCRL_COLD_DUAL, and mpcrl_get_iterate's bnd describes the last solve that did store them
                            (mpcrl_get_lagrangian stays current).  For callers that start every solve cold (a replay batch, the benchmark) or only ever
                            warm-start the primal iterate.  Honoured by the linear-system solve kernel (lq_solve_kernel); the
                            other kernels ignore it (they store, which is always allowed) */
    MPCRL_EXACT_QP = 64  /* ABI 120, TEST-ONLY, full SQP only (not with MPCRL_RTI, not with MPCRL_LINEAR_SPL=1: MPCRL_E_ARG): every QP of
                            the SQP is solved to the tight interior-point tolerance from a cold interior-point start, fixed
                            fraction to the boundary, no predictor-only steps — what acados + HPIPM do with the reference's
                            options (config/cartpole.yaml:8-14, chain_mass/ocp_utils.py:308-312) and what the oracle's frozen
                            ORACLE_EXACT mode does.  Cartpole: the plain launch shape of a separate kernel instantiation; chain of
                            masses and the linear-system kernel: a wave-uniform switch at the SQP level.  tests/test_gpu_fullsize.py
                            holds the shipped (inexact-SQP) iteration to it at 1e-6 on the benchmark's inputs.  Several times
                            slower: never use it to time */
};
enum { MPCRL_E_ARG = -1, MPCRL_E_MODEL = -2, MPCRL_E_HIP = -3, MPCRL_E_NOMEM = -4 };

#define MPCRL_NO_BOUND 1e30 /* |bound| >= 1e29 means "absent" */

typedef struct {
    int32_t model;     /* MPCRL_MODEL_* */
    int32_t N;         /* horizon */
    int32_t nx, nu;    /* must match the model */
    int32_t np;        /* length of the full parameter vector p (reference order, nlp.py:969-989) */
    int32_t cost_kind; /* MPCRL_COST_* */
    double dT;         /* tf / N  (nlp.py:1163-1164) */
    double gamma;      /* discount factor */
    double h;          /* RK4 sub-step */
    int32_t rk_steps;  /* RK4 steps per shooting interval */
    double tol;        /* NLP residual tolerance (acados default 1e-6; chain 1e-5) */
    int32_t max_iter;  /* SQP iterations (nlp_solver_max_iter) */
    /* box bounds, HOST pointers, stage-vector order v = [u; x] */
    const double *lb0, *ub0; /* nu      : controls at stage 0 */
    const double *lb, *ub;   /* nu + nx : stages 1..N-1 */
    const double *lbe, *ube; /* nx      : stage N */
    const int32_t *soft;     /* nu + nx : 1 = L1-soft bound (idxsbx), may be NULL */
    const double *zl, *zu;   /* nu + nx : L1 weights of soft bounds, may be NULL */
    const double *consts;    /* model constants, HOST pointer (layout: DESIGN.md "model constants") */
    int32_t n_consts;
} MpcrlProblemSpec;

typedef struct MpcrlSolver *mpcrl_handle;

/* Creates a solver for `batch` independent OCP instances on HIP device `device`.  Copies the spec. */
int mpcrl_create(const MpcrlProblemSpec *spec, int batch, int device, mpcrl_handle *out);
int mpcrl_destroy(mpcrl_handle h);

/* theta: device pointer to np doubles (per_instance = 0, shared) or batch*np (per_instance = 1). */
int mpcrl_set_theta(mpcrl_handle h, const double *theta, int n_theta, int per_instance, void *stream);
int mpcrl_set_gamma(mpcrl_handle h, double gamma);
int mpcrl_set_options(mpcrl_handle h, double tol, int max_iter);

/* Opt-in divergence exit (not a reference behaviour: the reference runs full-step SQP without globalisation until
 * nlp_solver_max_iter, config/cartpole.yaml:12-14, rlmpc/mpc/common/mpc.py:42,81-83 — one instance at a time, where a diverging
 * solve costs only itself; in a batch the instances of a wavefront run in lock step and ONE diverging instance keeps its
 * wavefront iterating to max_iter).  Every `window` SQP iterations of an instance the best NLP residual seen so far must have
 * dropped below `factor` times its value at the previous check, else the instance ends with status 2 (as at max_iter) and its
 * lanes are free.  window = 0 (default): off.  1 <= window <= 255, 0 < factor <= 1.  Instances that converge with the rule on
 * return bit for bit what they return with it off.  All three model families (the chain of masses since ABI 110: there an instance
 * has a wavefront — a SIMD — to itself, and a diverging one holds it to max_iter). */
int mpcrl_set_exit_rule(mpcrl_handle h, int window, double factor);

/* Scheduling hint (no effect on results): perm[B] int32 on the device, a permutation of 0..B-1 — slot i of a launch works on
 * instance perm[i], so that instances expected to need similar iteration counts share a wavefront. NULL = identity. */
int mpcrl_set_order(mpcrl_handle h, const int32_t *perm, void *stream);
/* Builds that permutation on the device from the initial states themselves (x0: [B, nx] device): the batch ordered along the
 * coordinate of x0 with the largest spread.  One small kernel; batch <= 8192, else MPCRL_E_ARG (use mpcrl_set_order).  A no-op that
 * clears the order where every instance has a wavefront to itself (chain of masses; horizons of more than 31 stages). */
int mpcrl_auto_order(mpcrl_handle h, const double *x0, void *stream);
/* 1 if an mpcrl_solve with these flags would use the time-sliced launch (whose wavefronts take one instance from each quarter of the
 * batch: a packing order buys nothing there and the caller can skip building one), 0 if not, < 0 on misuse.  stream = the stream the
 * solve will be launched on: while it is being captured into a graph the answer is the shape such a launch takes (the one preferred
 * so far — nothing is timed inside a capture), not the shape of a probe call. */
int mpcrl_query_time_sliced(mpcrl_handle h, int flags, void *stream);
/* Launch shape of the cartpole / linear-system solve kernel for non-RTI solves (no effect on results: the two shapes return the same
 * bits).  mode 0 (default) = automatic: where the batch size makes the time-sliced shape a candidate (it saves a round of wavefronts
 * on the device's SIMDs) the handle times the two shapes against each other on the caller's own batches — calls 1 and 2 of every 64
 * are the probes, read back without waiting; cold calls and warm calls (stored iterates) are timed separately — and uses the plain
 * one where its kernel is more than 20 % faster: time-sliced on batches of one difficulty, plain on batches whose instances need very
 * different iteration counts (replay samples).  Inside a stream capture nothing is timed and the graph gets the shape preferred so
 * far.  mode 1 = time-sliced whenever legal, mode -1 = never (what the environment variable MPCRL_TIME_SLICE=1|0 sets at creation). */
int mpcrl_set_launch_mode(mpcrl_handle h, int mode);
/* The tuner's last probe times in ms (-1 = not measured yet) for cold (warm = 0) or warm (warm = 1) calls; returns 0 if it currently
 * prefers the time-sliced shape, 1 if the plain one, < 0 on misuse.  Never waits for the device. */
int mpcrl_get_launch_times(mpcrl_handle h, int warm, double *sliced_ms, double *plain_ms);

/* Box bounds after creation — ocp_solver.constraints_set(stage, "lbu"|"ubu"|"lbx"|"ubx", v) (rlmpc/mpc/common/mpc.py:72-73,87-88).
 * HOST pointers, stage-vector order v = [u; x]; |bound| >= 1e29 = absent.  which: U0 = controls of stage 0 (nu values),
 * STAGE = stages 1..N-1 (nu + nx; a coordinate created as a soft bound must keep both sides), TERMINAL = stage N (nx).
 * The pin lbu_0 = ubu_0 = u0 of Q(s,a) is the u0_fixed argument of mpcrl_solve (per instance), not this call. */
enum { MPCRL_BOUNDS_U0 = 0, MPCRL_BOUNDS_STAGE = 1, MPCRL_BOUNDS_TERMINAL = 2 };
int mpcrl_set_bounds(mpcrl_handle h, int which, const double *lb, const double *ub);

/* Cold iterate: x_k := x0 for all k (x0: [B, nx] device, or NULL: 0), u := 0, all multipliers 0, slacks 1; the next solve
 * starts cold (interior point from its default point). */
int mpcrl_reset(mpcrl_handle h, const double *x0, void *stream);

/* Per-instance MPCRL_COLD for the NEXT mpcrl_solve only: mask [B] int32 on the device, non-zero = that instance ignores its stored
 * iterate and starts from x_k = x0, u = 0, multipliers 0 (the reference's per-environment `mpc.reset(obs)` when an episode ends,
 * scripts/cartpole_mpc_as_td3_agent_closed_loop.py:62-64, for a batch in which only some environments ended).  NULL clears it. */
int mpcrl_set_cold_mask(mpcrl_handle h, const int32_t *mask, void *stream);

/* Solves all instances.
 *   x0        [B, nx]            initial states
 *   u0_fixed  [B, nu] or NULL    NULL: policy / V mode; else Q(s,a) mode (lbu_0 = ubu_0 = u0)
 *   u0_out    [B, nu]            u_0*
 *   V         [B]                optimal cost (V or Q)
 *   dV_dp     [B, np] or NULL    needs MPCRL_SENS_V
 *   dpi_dp    [B, nu, np] or NULL needs MPCRL_SENS_PI
 *   status    [B] int32          0 success, 1 NaN (a residual or the cost is not finite, e.g. a NaN in x0), 2 max-iter, 4 QP failure
 * The sensitivity rows are written in full by every call: zeros for the entries of p without a gradient, for instances whose
 * status is neither 0 nor 2, and for du0* / dp in Q mode; NaN where the exact-Hessian KKT matrix is not positive definite.
 *   iters     [B, 2] int32 or NULL   SQP iterations, total interior-point iterations
 */
int mpcrl_solve(mpcrl_handle h, const double *x0, const double *u0_fixed, int flags, double *u0_out, double *V,
                double *dV_dp, double *dpi_dp, int32_t *status, int32_t *iters, void *stream);

/* Iterate access (device arrays; any may be NULL).
 *   x [B, N+1, nx]; u [B, N, nu]; pi [B, N, nx] (pi[k] multiplies F(x_k,u_k) - x_{k+1}, nlp.py:827,1180)
 *   bnd [B, 10, N+1, nu+nx]: lam_l, lam_u, t_l, t_u, s_l, s_u, lam_sl, lam_su, t_sl, t_su
 *   res [B, 4]: stationarity, equality, inequality, complementarity residual of the last solve */
int mpcrl_get_iterate(mpcrl_handle h, double *x, double *u, double *pi, double *bnd, double *res, void *stream);
int mpcrl_set_iterate(mpcrl_handle h, const double *x, const double *u, const double *pi, const double *bnd, void *stream);
/* ABI 120.  The same moves with a ROW INDEX, for tables of iterates that hold more (or other) rows than the handle has instances —
 * a replay buffer that keeps, next to every transition, the iterate the roll-out policy's solve ended with (mpc4rl_amd/td3.py,
 * replay_iterates; the reference's SB3 loop keeps ONE solver and warm-starts every replay solve from the previous, unrelated sample:
 * rlmpc/td3/policies.py:186-213).  x [rows, (N+1) nx], u [rows, N nu], pi [rows, N nx], bnd [rows, 10 (N+1)(nu+nx)] device;
 * index [B] int64 device (NULL = identity).  get: table row index[i] := stored iterate of instance i (NULL arrays are skipped).
 * set: stored iterate of instance i := table row index[i]; bnd = NULL as in mpcrl_set_iterate (next solve: MPCRL_COLD_DUAL).
 * index[i] < 0: instance i is skipped (get: nothing written for it; set: it keeps its stored iterate — with bnd = NULL every
 * instance's bound planes are reset, skipped or not).  One launch each, no host synchronisation: capture-safe. */
int mpcrl_get_iterate_rows(mpcrl_handle h, double *x, double *u, double *pi, double *bnd, const int64_t *index, void *stream);
int mpcrl_set_iterate_rows(mpcrl_handle h, const double *x, const double *u, const double *pi, const double *bnd, const int64_t *index, void *stream);
/* L [B] device: the Lagrangian of the mirror NLP at the iterate the last solve returned, L = cost + pi'g + lam'h
 * (nlp.L, rlmpc/mpc/nlp.py:1180,1390; MPC.get_L, rlmpc/mpc/common/mpc.py:325-332). */
int mpcrl_get_lagrangian(mpcrl_handle h, double *L, void *stream);

/* K5, the local half of the one collective on the path: out[j] = sum_i weight[i] * grad[i*ld + j] (j < n), out[n] = sum_i weight[i],
 * out[n+1] = rows.  Replaces the per-sample Python accumulation of rlmpc/examples/linear_system_mpc_qlearning.py:203
 * (np.mean(np.vstack([LR * td[i] * dQ_dp[i, :] ...]))); the n+2 doubles are then all-reduced over the ranks (RCCL).
 * Device pointers; weight may be NULL (= 1).  No handle: the launch goes to the device that owns `out` (hipPointerGetAttributes),
 * whatever device is current in the calling thread, which is restored on return; MPCRL_E_ARG if `out` is not device memory. */
int mpcrl_weighted_grad_sum(const double *grad, int64_t ld, const double *weight, int rows, int n, double *out, void *stream);

/* K4, the batched cartpole swing-up environment (rlmpc/gym/continuous_cartpole/environment.py): B environments, one launch per call.
 *   par [9] HOST: gravity, masscart, masspole, length, force_mag, tau, x_threshold, theta_threshold, max_episode_steps
 *   state [B, 4], steps [B] int64: the environments (device, updated in place)
 * step  — action [B] in [-1, 1]: explicit Euler step (environment.py:105-134), obs [B, 4] (may be NULL) = new state, reward [B] =
 *         x^2 + theta^2 of the new state (:193-194), terminated [B] uint8 = the terminal box (:136-146), truncated [B] uint8 =
 *         steps >= max_episode_steps (gymnasium TimeLimit).
 * reset — environments with mask[i] != 0 (mask NULL: all) restart at (0, 0, (0.9 + 0.2 u01[i]) pi, 0) (:178-180), steps = 0;
 *         u01 [B] uniform [0, 1) numbers drawn by the caller; obs [B, 4] (may be NULL) = the state of EVERY environment after it.
 * obs_f32 != 0: obs is float (what the reference's gymnasium environments return, environment.py:166,186), else double; the state
 * is always double (the reference's numpy state).  No handle: the three environment calls launch on the device that owns `state`
 * (hipPointerGetAttributes), whatever device is current in the calling thread; device or managed memory (hipMalloc / hipMallocManaged),
 * MPCRL_E_ARG for anything else (host-registered or unknown pointers). */
int mpcrl_env_cartpole_step(const double *par, int B, double *state, int64_t *steps, const double *action, void *obs, int obs_f32,
                            double *reward, uint8_t *terminated, uint8_t *truncated, void *stream);
int mpcrl_env_cartpole_reset(int B, double *state, int64_t *steps, const uint8_t *mask, const double *u01, void *obs, int obs_f32,
                             void *stream);
/* The linear-system environment (rlmpc/gym/linear_system/environment.py:28-58), B environments in one launch:
 *   par [12] HOST: A (row-major 2x2), B (2), lb_noise, ub_noise, min_observation (2), max_observation (2)
 *   state [B, 2] (device, updated in place): s+ = A s + B a + [lb_noise + (ub_noise - lb_noise) u01, 0]; action [B]; u01 [B] uniform
 *   numbers of the caller; obs [B, 2] (may be NULL) = new state; cost [B] = 1/2 s's + 1/2 a'a + 100 per violated side of the box. */
int mpcrl_env_linear_step(const double *par, int B, double *state, const double *action, const double *u01, void *obs, int obs_f32,
                          double *cost, void *stream);

/* ABI 120.  The actor's output stage for a batch, one launch: what Actor.forward (rlmpc/td3/policies.py:186-213) and MPC.scale_action
 * (rlmpc/mpc/common/mpc.py:290-301) do per observation, plus TD3's exploration / target-policy noise.  All device pointers:
 *   u0 [B, nu] (mpcrl_solve's output), status [B]; noise [B, nu] float standard-normal draws of the caller or NULL; lo, hi [nu] = lbu, ubu;
 *   ok_i = status_i == 0 (accept_status2: or == 2) and u0_i finite;  a_i = ok_i ? 2 (u0_i - lo) / (hi - lo) - 1 : 0   (scale = 0: u0_i);
 *   noise given: a_i = clip(a_i + clip(sigma noise_i, +-noise_clip), -1, 1)   (noise_clip <= 0: the noise is not clipped);
 *   action [B, nu] float; ok [B] uint8, may be NULL.  Handle-less like the environment kernels (launched on the device that owns `action`). */
int mpcrl_policy_action(const double *u0, const int32_t *status, const float *noise, const double *lo, const double *hi, int B, int nu, int scale,
                        double sigma, double noise_clip, int accept_status2, float *action, uint8_t *ok, void *stream);

/* ABI 130.  The critic side of one TD3 update, two launches (critic_kernel.hpp): what stable_baselines3's TD3.train (the reference's
 * learner, pyproject.toml:10, around the critics of rlmpc/td3/policies.py:47-122) does between the target actor's action and the critic
 * optimiser's step.  Critics: n_critics (1 or 2) MLPs [obs | action] -> 64 -> 64 -> 1 with ReLU, float; `params` / `params_target` hold them
 * back to back in torch's parameter order, each W1 [64][nx + nu] | b1 [64] | W2 [64][64] | b2 [64] | W3 [64] | b3 [1] (row-major
 * [out][in], what nn.Linear.weight is).  All device pointers:
 *   rows [B][row_stride] float: obs (nx) | next obs (nx) | action (nu) | reward | done (the packed replay row), row_stride >= 2 nx + nu + 2;
 *   a_next [B][nu] float: the target policy's action at the next obs;  ok_u [B] uint8 or NULL: 0 = leave the transition out;
 *   ok_b   = ok_u[b] and the 2 nx + nu + 2 entries of the row and a_next[b] are finite;
 *   y_b    = reward + gamma (1 - done) min_c Q'_c(next obs, a_next);   e_cb = ok_b ? Q_c(obs, action) - y_b : 0;
 *   loss   = sum_c sum_b e_cb^2 / max(1, sum_b ok_b)   -> loss_out [1] float (may be NULL);
 *   grad   [n_params] DOUBLE = out_scale * d loss / d params, in the order of `params` (n_params = n_critics (64 (nx + nu) + 4289));
 *   ok_out [B] uint8 or NULL: ok_b.
 *   workspace: mpcrl_critic_workspace_bytes(B, nx, nu, n_critics) bytes of device memory owned by the caller.
 * The partial sums are reduced in a fixed order (no atomics): the same inputs give the same bits.  nx + nu <= 64, hidden width 64 only
 * (MPCRL_E_ARG otherwise).  Handle-less (launched on the device that owns `grad`). */
int64_t mpcrl_critic_workspace_bytes(int B, int nx, int nu, int n_critics);
int mpcrl_critic_td_grad(const float *rows, int row_stride, int B, int nx, int nu, const float *a_next, const uint8_t *ok_u, const float *params,
                         const float *params_target, int n_critics, double gamma, double out_scale, void *workspace, double *grad, float *loss_out,
                         uint8_t *ok_out, void *stream);

/* ABI 130.  dQ_1/da at (obs_b, act_b) for the deterministic policy gradient (autograd of ContinuousCritic.q1_forward,
 * rlmpc/td3/policies.py:68-76), one launch: obs [B][obs_stride] float (first nx entries), act [B][nu] float, ok [B] uint8 or NULL,
 * params: the FIRST critic in the layout above; dq_da [B][nu] float, 0 where ok[b] = 0 or an input is not finite; ok_out [B] uint8 or NULL:
 * 1 where the row was used. */
int mpcrl_critic_dq_da(const float *obs, int obs_stride, int B, int nx, int nu, const float *act, const uint8_t *ok, const float *params,
                       float *dq_da, uint8_t *ok_out, void *stream);

/* ABI 130.  A replay batch in one launch (replay_kernel.hpp): the sampled transitions gathered, their states widened for the two replay
 * solves, and — with iter_ok — the rows of the caller's iterate tables (mpcrl_set_iterate_rows) those solves start from.  Device pointers:
 *   table [cap * E][row_len] float: obs (nx) | next obs (nx) | action | reward | done (last entry), row = step * E + env;
 *   idx [B] int64: the sampled rows (the caller draws them);  pos_t [1] int64: the slot the roll-out writes next;  steps: slots written;
 *   rows [B][row_len] float = table[idx];  obs64 / nxt64 [B][nx] double;
 *   iter_ok [cap * E] uint8 or NULL (then the four outputs below are not written):
 *     row_s = idx;  row_n = the row of the NEXT step of the same environment if done == 0 and that slot is written and is not pos_t,
 *     else idx;  cold_s / cold_n [B] int32 = 1 where iter_ok[row] == 0 (mpcrl_set_cold_mask takes them as they are).
 *   exclude_pos != 0 (a full table only): slot pos_t is being written while the batch is drawn (a roll-out running beside the update):
 *     idx [B] in [0, (cap - 1) E) then counts the rows of the other slots, slot (pos_t + 1 + idx / E) % cap. */
int mpcrl_replay_sample(const float *table, int row_len, int nx, int E, int cap, int steps, const int64_t *idx, int B, const int64_t *pos_t,
                        int exclude_pos, const uint8_t *iter_ok, float *rows, double *obs64, double *nxt64, int64_t *row_s, int32_t *cold_s,
                        int64_t *row_n, int32_t *cold_n, void *stream);

/* ABI 130.  The contraction of the deterministic policy gradient, one launch: out[p] = sum_b ok_b sum_u dq_da[b][u] chain_u dpi_dp[b][u][p]
 * (p < n_p), out[n_p] = sum_b ok_b, chain_u = 2 / (hi_u - lo_u) with scale != 0 (the derivative of MPC.scale_action,
 * rlmpc/mpc/common/mpc.py:290-301), else 1.  dq_da [B][nu] float (mpcrl_critic_dq_da), ok [B] uint8 or NULL, dpi_dp [B][nu][n_p] double
 * (mpcrl_solve's output; entries of rows that are left in are read as nan_to_num would), lo / hi [nu] double, out [n_p + 1] double.
 * workspace: mpcrl_dpg_workspace_bytes(B, n_p) bytes of device memory, ZERO before the first call (the call leaves its counter zero);
 * fixed summation order: the same inputs give the same bits.  nu <= 8. */
int64_t mpcrl_dpg_workspace_bytes(int B, int n_p);
int mpcrl_dpg_grad(const float *dq_da, const uint8_t *ok, const double *dpi_dp, int B, int nu, int n_p, const double *lo, const double *hi, int scale,
                   void *workspace, double *out, void *stream);

/* ABI 130.  The roll-out side of one TD3 step after the policy's solve, one launch (td3_kernel.hpp), cartpole environment, nu = 1: the
 * actor's output stage with exploration noise (mpcrl_policy_action with accept_status2), mpcrl_env_cartpole_step, the replay row
 * [obs | next obs | action | reward_scale * reward | terminated] (float, 11 entries) written at slot pos[0] of table [cap][E][11], the flag
 * of the stored iterate (iter_ok[pos][env] = solve converged and u0 finite; may be NULL), the statistics (stats[0..2] += sum of rewards,
 * converged solves, episodes ended), the reset of the environments that ended (mpcrl_env_cartpole_reset with u01), the next observation
 * obs [E][4] double (in: the observation of this solve) and ended [E] int32 (the cold mask of the next solve).  The last workgroup
 * advances pos[0] to (pos + 1) % cap; iter_rows [E] int64 (may be NULL) receives the rows of this step in the caller's iterate tables,
 * pos * E + env (what mpcrl_get_iterate_rows is then called with).
 * par: the nine doubles of mpcrl_env_cartpole_step.  workspace: 16 + 24 * ceil(E / 256) bytes, ZERO before the first call.  The
 * arithmetic is that of the three kernels it stands for (shared device functions): a loop switched to it reproduces its numbers. */
int mpcrl_td3_cartpole_collect(const double *par, int E, double *state, int64_t *steps, const double *u0, const int32_t *status, const float *eps,
                               const double *u01, double lo, double hi, int scale, double sigma, double *obs, int32_t *ended, float *table, int cap,
                               double reward_scale, int64_t *pos, uint8_t *iter_ok, int64_t *iter_rows, double *stats, void *workspace, void *stream);

/* ABI 130.  The policy half of a TD3 update after the collective, one launch: msg [n_theta + 1] double = the all-reduced theta-gradient
 * sum and sample count;  step = lr * mask * msg / max(1, count) -> step_out;  theta += step;  theta_target = (1 - tau) theta_target +
 * tau theta;  crit_target = (1 - tau) crit_target + tau crit (float [n_crit], the flat critic parameters; n_crit may be 0). */
int mpcrl_td3_policy_post(const double *msg, int n_theta, double lr, const double *mask, double tau, double *theta, double *theta_target,
                          double *step_out, const float *crit, float *crit_target, int n_crit, void *stream);

/* Bytes of device memory held by the handle; library version (MPCRL_ABI_VERSION of the header it was built from). */
int64_t mpcrl_workspace_bytes(mpcrl_handle h);
int mpcrl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCRL_H */
