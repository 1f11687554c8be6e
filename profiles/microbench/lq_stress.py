"""lq_solve_kernel vs the one-stage-per-lane kernels on variations the parity tests do not sweep: discount factors, changed bounds
(missing rows, asymmetric, terminal), exit rule, parameters with offsets, several horizons and batch sizes, many seeds."""
import os, sys, itertools, numpy as np, torch
sys.path.insert(0, '.')
from mpc4rl_amd import MPCBatch, linear_system_ocp, _lib
worst = {}
stat = {"bad_old": 0, "bad_new": 0, "status_differs": 0, "sqp_differs": 0, "ipm_differs_by_more_than_1": 0, "ipm_differs_by_1": 0, "instances": 0}
def cmp(tag, ra, rb):
    n = len(ra.status)
    stat["instances"] += n
    stat["bad_old"] += int((ra.status != 0).sum()); stat["bad_new"] += int((rb.status != 0).sum())
    stat["status_differs"] += int((ra.status != rb.status).sum())
    stat["sqp_differs"] += int((ra.iters[:, 0] != rb.iters[:, 0]).sum())
    dit = (ra.iters[:, 1] - rb.iters[:, 1]).abs()
    same_sqp = ra.iters[:, 0] == rb.iters[:, 0]
    stat["ipm_differs_by_more_than_1"] += int(((dit > 1) & same_sqp).sum()); stat["ipm_differs_by_1"] += int(((dit == 1) & same_sqp).sum())
    if tag[-1] != 0 and (int(((dit > 1) & same_sqp).sum()) or int((ra.status != rb.status).sum())): print(tag, "warm call: status/ipm differences", int((ra.status != rb.status).sum()), int(((dit > 1) & same_sqp).sum()))
    if tag[-1] == 0: assert torch.equal(ra.status, rb.status) and bool(same_sqp.all()) and int(dit.max()) <= 1, (tag, "cold call")
    ok = (ra.status == 0) & (rb.status == 0)
    for f in ('u0', 'V', 'dV_dp', 'dpi_dp'):
        a, b = getattr(ra, f), getattr(rb, f)
        if a is None: continue
        a, b = torch.nan_to_num(a[ok]), torch.nan_to_num(b[ok])
        if a.numel() == 0: continue
        e = float((a - b).abs().max() / max(1.0, float(a.abs().max())))
        worst[f] = max(worst.get(f, 0.0), e)
    return 0
n_cases = flips = 0
for seed, (N, B), gamma in itertools.product(range(3), [(40, 257), (40, 64), (13, 40), (30, 33), (45, 48), (55, 20)], (0.99, 0.9, 1.0)):
    rng = np.random.default_rng(1000 * seed + N + B)
    ocp = linear_system_ocp(discount_factor=gamma, N=N)
    x0 = np.column_stack([rng.uniform(0.05, 0.95, B), rng.uniform(-0.8, 0.8, B)])
    theta = np.tile(ocp.p0, (B, 1))
    theta[:, :6] *= rng.uniform(0.95, 1.05, (B, 6))
    theta[:, 6:8] = rng.normal(0, 0.01, (B, 2))
    theta[:, 8] = rng.normal(0, 0.1, B)
    theta[:, 9:] = rng.normal(0, 0.05, (B, 3))
    variant = seed % 3
    u0 = torch.as_tensor(rng.uniform(-0.5, 0.5, (B, 1)), device='cuda')
    res = {}
    for spl in ('1', '3'):
        os.environ['MPCRL_LINEAR_SPL'] = spl
        m = MPCBatch(ocp, B)
        m.set_theta(torch.as_tensor(theta))
        if variant == 1:      # no lower bound on the second state, tighter controls, asymmetric terminal box
            m.set_bounds(_lib.BOUNDS_U0, [-0.6], [0.8])
            m.set_bounds(_lib.BOUNDS_STAGE, [-0.6, 0.0, -1e30], [0.8, 1.0, 0.7])
            m.set_bounds(_lib.BOUNDS_TERMINAL, [0.0, -0.5], [1.0, 1e30])
        if variant == 2:
            m.set_exit_rule(5, 0.5)
            m.set_options(tol=1e-8, max_iter=30)
        seq = [m.solve(x0, sens_v=True, sens_pi=True, cold=True)]
        seq.append(m.solve(x0 + 0.01, sens_v=True, sens_pi=True))
        seq.append(m.solve(x0 + 0.01, u0, sens_v=True))
        res[spl] = seq
    for n, (ra, rb) in enumerate(zip(res['1'], res['3'])):
        flips += cmp((seed, N, B, gamma, variant, n), ra, rb)
        n_cases += 1
print("cases", n_cases, stat, "worst relative differences over instances both solve", worst)
