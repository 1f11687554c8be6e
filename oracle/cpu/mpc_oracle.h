/* C interface of the CPU oracle port ("port" kind CPU baseline).  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The struct is the oracle's own; the product's C ABI lives in include/mpcrl.h. */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_MODEL_CARTPOLE = 0, ORACLE_MODEL_LINEAR = 1, ORACLE_MODEL_CHAIN = 2 };
enum { ORACLE_COST_NLS = 0, ORACLE_COST_EXTERNAL = 1 };
enum {
    ORACLE_SENS_V = 1,
    ORACLE_SENS_PI = 2,
    ORACLE_WARM = 4,
    /* EXACT-QP mode (frozen; what acados + HPIPM do with the reference's options, config/cartpole.yaml:8-14): every QP is
     * solved to the tight tolerances from a cold interior-point start with the fixed fraction to the boundary 0.995 — no
     * inexact-SQP forcing term, no interior-point warm start, no adaptive step rule.  The product is tuned against THIS mode
     * (tests/test_exact_vs_inexact.py); its constants are not to be touched when the product's iteration is tuned. */
    ORACLE_EXACT = 8,
    ORACLE_RTI = 16 /* one SQP iteration from the given iterate (the product's MPCRL_RTI; not a reference mode) */
};


typedef struct {
    int model;          /* ORACLE_MODEL_* */
    int N, nx, nu, np;  /* horizon, dims, length of the full parameter vector p */
    int cost_kind;      /* ORACLE_COST_*: how c_k is built from dT and gamma (nlp.py:1044-1055 / 1083-1091) */
    double dT, gamma;
    double h;           /* RK4 sub-step */
    int rk_steps;       /* RK4 steps per shooting interval */
    double tol;         /* NLP residual tolerance */
    int max_iter;       /* SQP iterations */
    /* box bounds in stage-vector order v = [u; x]; +-1e30 = absent */
    const double *lb0, *ub0;   /* nu      : stage 0 controls */
    const double *lb, *ub;     /* nu + nx : stages 1..N-1 */
    const double *lbe, *ube;   /* nx      : stage N */
    const int *soft;           /* nu + nx : 1 = L1-soft bound on stages 1..N-1 */
    const double *zl, *zu;     /* nu + nx : L1 weights of the soft bounds */
    const double *consts;      /* model constants (see mpc_oracle.cpp, per model) */
    int n_consts;
    /* the product's opt-in divergence exit (mpcrl_set_exit_rule): every exit_window SQP iterations the best NLP residual so far must
     * be below exit_factor x its value at the previous check, else status 2; 0 = off (the reference's behaviour) */
    int exit_window;
    double exit_factor;
} OracleSpec;

/* Solves B independent OCPs.  All arrays are host, row-major, double.
 * p: np values (shared) or B*np (p_per_instance).  u0fix NULL = policy mode, else Q-mode (mpc.py:52-96).
 * X,U,PI: iterate out (in as well with ORACLE_WARM).  BND: B x 10 x (N+1) x (nu+nx):
 *   lam_l, lam_u, t_l, t_u, s_l, s_u, lam_sl, lam_su, t_sl, t_su   (may be NULL; in+out with ORACLE_WARM)
 * dV: B x np, dpi: B x nu x np.  res: B x 4 (stat, eq, ineq, comp).  Returns 0, or <0 on bad arguments. */
int mpc_oracle_solve(const OracleSpec *sp, int B, const double *x0, const double *u0fix, const double *p,
                     int p_per_instance, int flags, double *X, double *U, double *PI, double *BND, double *u0_out,
                     double *V, double *dV, double *dpi, int *status, int *sqp_iter, int *ipm_iter, double *res,
                     int nthreads);

#ifdef __cplusplus
}
#endif
#endif
